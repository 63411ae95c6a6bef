"""CPU oracle for the ViT encoder hot path (TEST INFRASTRUCTURE ONLY).

This package is a plain-PyTorch, CPU, fp32 restatement of the reference's
algorithm (``/root/reference/vit_pytorch_robust/simple_vit.py``, ``vit.py``,
``utils.py:1025-1037``, ``mae.py``) written independently of the product code
under ``noise_robust_vit_amd/``.  It exists so that parity tests can compare
the HIP path against the reference's arithmetic on a box that does not hold
the reference.

Rules (enforced by ``tests/test_host_logic.py::test_product_never_imports_the_oracle_or_reads_the_reference``):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here;
  * nothing under ``noise_robust_vit_amd/`` imports it, and the product path
    raises when the HIP library is missing rather than falling back to this.

Pinning: ``tests/golden/*.npz`` hold inputs/weights/outputs produced by
importing the *reference itself* in the development container
(``tests/golden/gen_golden.py``); ``tests/test_oracle_golden.py`` checks this
restatement against them (SimpleViT path, softmax and Sinkhorn attention).
The VisionTransformer path has no runnable reference forward (SURVEY.md §0,
§8c) -- its oracle is pinned against ``torch.nn.MultiheadAttention`` /
``torch.nn.LayerNorm`` instead and is documented as "parity unpinned by the
reference".
"""
