"""CPU oracle for the polydet hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A plain torch-fp32 / numpy restatement of the reference's algorithm for every
row of SURVEY.md section 8(a).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this package; nothing under
`centerpoly_amd/` does, and the product path raises when the HIP library is
missing instead of falling back to anything here.

Pinning (tests/test_oracle_golden.py):
  * decode / losses / hourglass / DLA wiring / post-process are pinned against
    golden vectors produced by importing the reference's own Python on CPU
    (tools/gen_golden.py -> tests/golden/*.npz);
  * the Weiler-Atherton IoU term is additionally pinned by the known answers of
    SURVEY.md section 4 (measured with the reference code);
  * DCNv2 is a third-party dependency ABSENT from /root/reference
    (CharlesShang/DCNv2, no version pinned: README.md:57, NOTICE:146-154).
    oracle/dcn.py restates its published algorithm; the reference holds no test
    or fixture for it, so **DCN parity is unpinned by the reference** and is
    anchored on the zero-offset known answer, finite differences and the call
    site contract (src/lib/models/networks/pose_dla_dcn.py:16,354).
"""
