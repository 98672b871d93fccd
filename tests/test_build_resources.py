"""The streaming kernels must not spill: the compiler's own resource report of the product build (written by longspec_amd.build
next to each object) is part of the test suite.  A spill in these kernels is silent -- results stay right -- and costs a factor:
in round 3 an edit of two integer divisions in the split arithmetic made attn_partial_ws_kernel spill 512 registers and run 3.4x
slower; a scratch reload also makes the compiler drain the hand-counted LDS-DMA look-ahead (s_waitcnt vmcnt(0))."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _usage(src):
    from longspec_amd import build
    build.build(verbose=False)
    with open(os.path.join(build.LIBDIR, src + ".usage.json")) as f:
        return json.load(f)


def test_attention_kernels_have_no_scratch():
    u = _usage("attn")
    names = [n for n in u if "attn_partial" in n or "attn_finish" in n]
    assert len(names) >= 12, names
    for n in names:
        assert u[n]["VGPRs Spill"] == 0 and u[n]["ScratchSize [bytes/lane]"] == 0, (n, u[n])
    for n in names:
        if "attn_partial" in n:
            assert u[n]["Occupancy [waves/SIMD]"] >= 2, (n, u[n])     # 8 waves per workgroup: two per SIMD


def test_gemm_kernels_have_no_scratch():
    u = _usage("gemm")
    names = [n for n in u if "skinny_gemm_kernel" in n]
    assert names
    for n in names:
        assert u[n]["VGPRs Spill"] == 0 and u[n]["ScratchSize [bytes/lane]"] == 0, (n, u[n])
