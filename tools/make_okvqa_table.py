#!/usr/bin/env python3
"""Writes licv-vqa_amd/icv_src/metrics/okvqa_manual_matches.json: the OK-VQA v1.1 stemming EXCEPTION TABLE (word -> stem pairs found
by comparing the dataset's `raw_answers` with its `answers`) as data, read out of the reference checkout's
icv_src/metrics/okvqa_utils.py with `ast.literal_eval` (the module itself cannot be imported here: nltk is not installed).
Only the dictionary's key/value strings are written - no source text.  Run in the build container:

    python tools/make_okvqa_table.py [/root/reference]
"""
import ast
import json
import sys
from pathlib import Path

ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference") / "icv_src" / "metrics" / "okvqa_utils.py"
tree = ast.parse(ref.read_text())
table = None
for node in tree.body:
    if isinstance(node, ast.Assign) and any(getattr(t, "id", None) == "_MANUAL_MATCHES" for t in node.targets):
        table = ast.literal_eval(node.value)
assert isinstance(table, dict) and len(table) > 100, "exception table not found"
out = Path(__file__).resolve().parents[1] / "licv-vqa_amd" / "icv_src" / "metrics" / "okvqa_manual_matches.json"
out.write_text(json.dumps(table, indent=0, sort_keys=True, ensure_ascii=False) + "\n")
print(f"{len(table)} entries -> {out}")
