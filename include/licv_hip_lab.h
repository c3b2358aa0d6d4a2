/* liblicv_hip_lab.so - measurement code that is NOT part of the product library: GEMM kernel variants that were built, measured and
 * not adopted (csrc/lab/gemm_experiments.hip) and the roofline probes (csrc/lab/probe.hip).  Only tests/ and tools/ load it
 * (licv._lib.lab()).  It links against liblicv_hip.so and, when loaded, registers the experiments with licv_gemm_select's dispatch
 * (licv_lab_register in licv_hip.h).  Same conventions as licv_hip.h: 0 = ok, negative LICV_E_*; raw device pointers; explicit stream. */
#ifndef LICV_HIP_LAB_H
#define LICV_HIP_LAB_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* 1 once the library is loaded (its constructor has registered the experiments with liblicv_hip.so) */
int licv_lab_loaded(void);
/* A/B switch for the persistent experiment kernel's per-XCD start stagger (default off). */
int licv_gemm_stagger(int on);
/* roofline probe: `blocks` workgroups of 4 waves each issue iters*8 register-only v_mfma_f32_16x16x32_bf16 (16384 FLOP each) */
int licv_probe_mfma_loop(void* sink_f32, int blocks, int iters, void* stream);
/* semantics probe: one wave writes {a', b'} = v_permlane16_swap(a = lane, b = 100 + lane) to out[2*lane], out[2*lane+1] (uint32) */
int licv_probe_permlane16_swap(void* out_u32_128, void* stream);
/* roofline probe: stream a cold [N, K] bf16 matrix with 4-wave workgroups (16 rows x K/splits per wave) doing nothing with the data;
 * shape = bytes per row per instruction: 0 16 rows x 64 B (MFMA fragment order), 1 8 x 128 B, 2 2 x 512 B, 3 1 x 1 KB; depth = 16-byte
 * loads per lane in flight per register set (4, 8 or 16) */
int licv_probe_weight_stream(const void* W, int64_t ldw, int64_t N, int64_t K, int splits, int shape, int depth, void* sink_u32, void* stream);
/* the same stream as LDS-DMA pieces (8 rows x 128 B per instruction), `depth` (4 / 8 / 16 / 32 / 48) pieces outstanding per wave */
int licv_probe_lds_dma_stream(const void* W, int64_t ldw, int64_t N, int64_t K, int splits, int depth, void* stream);
/* per-CU operand bandwidth from an L2-resident buffer: every one of `blocks` workgroups reads the same `bytes` `reps` times; mode 0 =
 * buffer loads to VGPRs, 1 = LDS-DMA pieces, 2 = two waves each at once (each pair reads the whole buffer) */
int licv_probe_l2_ingest(const void* buf, int64_t bytes, int reps, int mode, int blocks, void* sink_u32, void* stream);
#ifdef __cplusplus
}
#endif
#endif
