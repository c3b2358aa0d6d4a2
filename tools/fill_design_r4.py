#!/usr/bin/env python3
"""Fills the placeholders of tools/design_r4_section.md from the committed round-4 profiles and splices the result into DESIGN.md as its
section 5 (the round logs of rounds 1 - 3 are kept below it as 5.5).  Build-container helper; run after copying the evidence run's files
into profiles/."""
import json
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
P = ROOT / "profiles"


def line(name):
    for l in (P / f"r04_bench_{name}.json.log").read_text().splitlines():
        if l.strip().startswith("{"):
            return json.loads(l)
    raise SystemExit(f"no JSON line in {name}")


def stats(fname):
    out = {}
    for l in (P / fname).read_text().splitlines():
        m = re.match(r"\| `(.+?)` \| (\d+) \| ([\d.]+) \| ([\d.]+)", l)
        if m:
            out[m.group(1)] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
    return out


W = [("headline", "headline `idefics9b_32shot_bs8` (configs[1])", "178.5 - 183.4"),
     ("idefics9b_train_bs8", "`idefics9b_train_bs8` (configs[2] on one GPU: teacher + student fwd / bwd + AdamW)", "207.5"),
     ("idefics9b_train_bs8_cached_vision", "`idefics9b_train_bs8_cached_vision` (f3: vision-feature cache warm)", "-"),
     ("idefics9b_train_bs8_cached_teacher", "`idefics9b_train_bs8_cached_teacher` (f3: + teacher rows cached)", "-"),
     ("idefics9b_student_bs8", "`idefics9b_student_bs8` (the 32-token hooked student pass)", "15.1"),
     ("idefics9b_generate_bs8", "`idefics9b_generate_bs8` (3 beams, 5 new tokens, hooks on)", "42.0 - 42.5"),
     ("idefics2_8b_1shot_bs8", "`idefics2_8b_1shot_bs8` (configs[3])", "46.6"),
     ("idefics2_8b_32shot_bs8", "`idefics2_8b_32shot_bs8`", "611.5"),
     ("idefics2_8b_32shot_fp8_bs8", "`idefics2_8b_32shot_fp8_bs8` (configs[4])", "455.0"),
     ("frontend_images_bs8", "`frontend_images_bs8` (uint8 images fed beside the forward)", "11.6 k images/s")]
rows, D = [], {}
for key, label, r3 in W:
    d = line(key)
    D[key] = d
    r = d.get("roofline", {})
    frac = r.get("frac")
    roof = f"{r.get('bound', '-').upper()} {frac:.3f}" if frac else "-"
    if key == "headline":
        roof += f" ({r['achieved']:.0f} TFLOP/s in-run), traffic {r['traffic']:.2f} GB / launch" if r.get("traffic") else f" ({r['achieved']:.0f} TFLOP/s in-run)"
    unit = "images/s" if key.startswith("frontend") else "questions/s"
    rows.append(f"| {label} | {d['value']:.1f}{' ' + unit if unit != 'questions/s' else ''} | {d['ms_per_step']:.1f} | {roof} | {r3} | `profiles/r04_bench_{key}.json.log` |")

hl = D["headline"]["roofline"]
ks = stats("r04_bench_headline_serial_kernel_stats.md")
gen = stats("r04_generate_9b_kernel_stats.md")
stu = stats("r04_student_9b_kernel_stats.md")
hook = D["headline"].get("hook_kernel", {})


def avg(st, prefix):
    for k, v in st.items():
        if k.startswith(prefix):
            return v[2]
    return float("nan")


def per_step(st, prefix, steps=7):
    c = sum(v[0] for k, v in st.items() if k.startswith(prefix))
    t = sum(v[1] for k, v in st.items() if k.startswith(prefix))
    return c / steps, t / steps


tr, cv, ct = D["idefics9b_train_bs8"], D["idefics9b_train_bs8_cached_vision"], D["idefics9b_train_bs8_cached_teacher"]
fp8, b32 = D["idefics2_8b_32shot_fp8_bs8"], D["idefics2_8b_32shot_bs8"]
cm, tm = per_step(stu, "gemm_bf16_mid_k")
cs, ts = per_step(gen, "gemm_bf16_skinny_k")
sub = {
    "@@BENCH_TABLE@@": "\n".join(rows),
    "@@TRAIN@@": f"{tr['ms_per_step']:.1f}", "@@TRAIN_QPS@@": f"{tr['value']:.1f}",
    "@@TRAIN_CV@@": f"{cv['ms_per_step']:.1f}", "@@TRAIN_CV_QPS@@": f"{cv['value']:.1f}",
    "@@TRAIN_CT@@": f"{ct['ms_per_step']:.1f}", "@@TRAIN_CT_QPS@@": f"{ct['value']:.1f}",
    "@@HL_TF@@": f"{hl['achieved']:.0f}", "@@HL_FRAC@@": f"{hl['frac']:.3f}", "@@HL_LAUNCHES@@": str(hl["launches_per_step"]),
    "@@HL_US@@": f"{hl['avg_launch_us']:.1f}", "@@HL_TRAFFIC@@": f"{hl['traffic']:.2f}" if hl.get("traffic") else "(see the PMC file)",
    "@@K0@@": f"{avg(ks, 'gemm_bf16_flow64_k<0>'):.1f}", "@@K5@@": f"{avg(ks, 'gemm_bf16_flow64_k<5>'):.1f}",
    "@@K4@@": f"{avg(ks, 'gemm_bf16_flow64_k<4>'):.1f}", "@@K1@@": f"{avg(ks, 'gemm_bf16_flow64_k<1>'):.1f}",
    "@@STU_MID@@": f"{cm:.0f} launches and {tm:.1f} ms per step ({D['idefics9b_student_bs8']['ms_per_step']:.1f} ms step): the text stack's M = 256 projections and the vision tower's M = 2056 ones",
    "@@GEN_SKINNY@@": f"{cs:.0f} launches and {ts:.1f} ms per generate ({D['idefics9b_generate_bs8']['ms_per_step']:.1f} ms): 4 decode steps x 161 projections, average {avg(gen, 'gemm_bf16_skinny_k<2'):.1f} us",
    "@@FP8_MS@@": f"{fp8['ms_per_step']:.1f}", "@@BF16_32_MS@@": f"{b32['ms_per_step']:.1f}", "@@FP8_RATIO@@": f"{fp8['ms_per_step'] / b32['ms_per_step']:.2f}",
    "@@ATTN_RES@@": f"{avg(ks, 'attn_resident_k'):.1f}", "@@ATTN_LM@@": f"{avg(ks, 'attn_fwd_k<128, 128, 1, 4, 1'):.1f}",
    "@@DEC_US@@": f"{avg(gen, 'decode_attn_k'):.1f}",
    "@@HOOK_US@@": f"{hook.get('avg_launch_us', float('nan')):.1f}", "@@HOOK_GBS@@": f"{hook.get('achieved', float('nan')):.0f}",
    "@@HOOK_FRAC@@": f"{hook.get('frac', float('nan')):.2f}",
    "@@GEN_MS@@": f"{D['idefics9b_generate_bs8']['ms_per_step']:.1f}", "@@GEN_FRAC@@": f"{D['idefics9b_generate_bs8']['roofline']['frac']:.3f}",
    "@@GEN_SKINNY_MS@@": f"{ts:.1f}",
    "@@STU_MS@@": f"{D['idefics9b_student_bs8']['ms_per_step']:.1f}", "@@STU_FRAC@@": f"{D['idefics9b_student_bs8']['roofline']['frac']:.3f}",
    "@@I2_1@@": f"{D['idefics2_8b_1shot_bs8']['ms_per_step']:.1f}", "@@I2_1_FRAC@@": f"{D['idefics2_8b_1shot_bs8']['roofline']['frac']:.2f}",
    "@@ROW_US@@": f"{avg(ks, 'add_rmsnorm_fwd_k<1'):.1f} / {avg(ks, 'rotary_fwd_k'):.1f} / {avg(ks, 'layernorm8_fwd_k'):.1f} us per call",
}
text = (ROOT / "tools" / "design_r4_section.md").read_text()
where = (ROOT / "tools" / "design_r4_where.md").read_text() if (ROOT / "tools" / "design_r4_where.md").exists() else "(to be written)"
for k, v in sub.items():
    where = where.replace(k, v)
sub["@@WHERE@@"] = where.strip()
for k, v in sub.items():
    text = text.replace(k, v)
left = re.findall(r"@@[A-Z0-9_]+@@", text)
assert not left, left
design = (ROOT / "DESIGN.md").read_text()
a = design.index("## 5. Measurement")
b = design.index("## 6. Multi-GPU")
old = design[a:b]
if "### 5.5 Round logs" in old:                          # already spliced once: keep its 5.5
    logs = old[old.index("### 5.5 Round logs"):]
    logs = logs[logs.index("\n") + 1:]
else:
    logs = old[len("## 5. Measurement"):].lstrip("\n")
    logs = logs.replace("### Where the time is now, and the leads that are left (end of round 3)", "#### Where the time was, and the leads that were left (end of round 3)")
    logs = logs.replace("### Where the time is now, and the leads that are left (end of round 2)", "#### Where the time was, and the leads that were left (end of round 2)")
    logs = logs.replace("### Round 3", "#### Round 3").replace("### Round 2", "#### Round 2").replace("### Round 1", "#### Round 1")
(ROOT / "DESIGN.md").write_text(design[:a] + text.rstrip("\n") + "\n\n" + logs.rstrip("\n") + "\n\n" + design[b:])
print("DESIGN.md section 5 rewritten;", len(rows), "bench lines")
