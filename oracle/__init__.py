"""CPU oracle for the contrastive hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a plain-PyTorch (CPU, fp32/fp64) *restatement* of the reference
algorithm for the one path this repo accelerates: encoders ->
projection -> L2 normalise -> all-pairs similarity -> symmetric InfoNCE (+ the
RAdam update that follows it).  It is written functionally over a flat
``{state_dict name: tensor}`` mapping, so it shares the reference's parameter
names but none of its module code.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- only as the checker / the timed CPU baseline.  The product
package ``multimodal_supernovae_amd`` never imports it and has no CPU fallback.

Parity status: PINNED.  Every function here is checked against golden vectors in
``tests/golden/*.npz`` that were produced by importing the real reference
(`/root/reference/src/{loss,transformer_utils,models_multimodal}.py`) in the build
container with ``tools/gen_golden.py`` (the reference's own test-suite holds no
vectors for this path -- SURVEY.md section 4).  Build-defined encoders (ViT, ResNet-18,
1-D CNN: not present in the reference) are "parity unpinned by the reference";
their oracle is the restatement in ``oracle/build_defined.py`` alone.
"""
