"""oracle/ -- CPU restatement of the reference's single-image inference hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py may import it, and only as the checker; nothing under hipie_amd/
imports it and the product fails loudly when its HIP library is missing.

The reference (berkeley-hipie/HIPIE, /root/reference) is Python on PyTorch, so the restatement is
plain fp32 PyTorch on the CPU -- functional code over a state_dict with the reference's key names,
each function citing the reference file:line it follows.  Third-party arithmetic the reference
calls (transformers' BertModel, nn.MultiheadAttention, F.grid_sample / F.interpolate, GroupNorm,
LayerNorm) is restated from its published definition or used through torch itself.

Pinning: the reference ships no tests for this path except the MSDeformAttn op check
(ops/test.py).  The oracle is therefore pinned against fixtures produced by running the
reference's OWN modules in the authoring container (tests/golden/gen_golden.py, through the import
shim tests/golden/ref_shim.py): per-kernel fixtures, stage fixtures and the full a22 output
dictionary of DDETRSegmUniDN.coco_inference on a tiny configuration.  tests/test_oracle_golden.py
checks every one of them on the CPU.
"""
