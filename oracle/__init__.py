"""CPU oracle for the L-ICV hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the arithmetic of the reference path
(ForJadeForest/LICV-VQA ``icv_src/`` + the HF Idefics model code it drives) so that the
hand-written HIP path can be checked against it.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it;
the product package (``licv-vqa_amd/``) never does and fails loudly when its HIP
library is missing.

Pinning: the reference ships no tests or golden vectors, so the oracle
is pinned by fixtures generated *in the build container* by ``tools/make_golden.py``
from the reference's own ``GlobalICVEncoder`` / ``LearnableICVInterventionLMM`` /
``VQAICVModule.forward`` code driving the installed ``transformers`` Idefics model
(tiny random-init configs).  ``tests/test_oracle_golden.py`` holds the oracle to those
fixtures (fp32 ≤ 1e-5, bf16 bit-for-bit on most tensors).
"""
