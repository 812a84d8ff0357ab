"""Static guard (no GPU): the kernels that keep the next tile in flight must not wait for it before the work it is meant to hide.

hipcc places `s_waitcnt vmcnt` statically, and three source patterns made it wait directly behind a prefetch in every kernel of
this repo that had one (docs/DESIGN_HISTORY.md "Round 3" table; tools/isa_wait_audit.py).  The fixes are source idioms a later edit can undo
without failing any numerical test -- the results do not change, only the overlap does -- so the compiled ISA is checked here:
between the requests of the next tile inside the main loop and the first `vmcnt` wait after them there must be the loop's
arithmetic (MFMAs, or the scan's state loop)."""
import os
import re
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")


def _body(src, name_part):
    import isa_wait_audit as audit
    path = audit.compile_s(os.path.join(audit.CSRC, src))
    found = [(n, b) for n, b in audit.kernels(path) if name_part in n]
    assert len(found) == 1, (name_part, [n for n, _ in found])
    return [x for x in found[0][1] if not x.endswith(":")]


def _after_last_request_group(ins, is_request, min_group):
    """index just behind the LAST run of >= min_group requests (the one inside the main loop; the first run is the prologue's)"""
    idx = [k for k, x in enumerate(ins) if is_request(x)]
    groups, cur = [], [idx[0]]
    for k in idx[1:]:
        if k - cur[-1] <= 120:
            cur.append(k)
        else:
            groups.append(cur)
            cur = [k]
    groups.append(cur)
    big = [g for g in groups if len(g) >= min_group]
    assert len(big) >= 2, f"expected a prologue group and an in-loop group of requests, found {[len(g) for g in groups]}"
    return big[-1][-1] + 1


def _work_before_wait(ins, start, is_work):
    n = 0
    for x in ins[start:]:
        if "vmcnt" in x:
            break
        n += 1 if is_work(x) else 0
    return n


def test_scan_bwd_dma_walk_requests_stay_in_flight_across_the_state_loop():
    ins = _body("scan_bwd.hip", "scan_bwd_kernelINS_6bf16_tELi8ELb1ELi16ELb0ELb1E")
    start = _after_last_request_group(ins, lambda x: "global_load_lds" in x, 5)
    valu = _work_before_wait(ins, start, lambda x: x.startswith("v_"))
    assert valu > 600, f"only {valu} VALU instructions between the DMA requests of the next chunk and the first vmcnt wait"
    # no compiler-visible vector load inside the chunk loop at all
    tail = ins[start:]
    end = next(k for k, x in enumerate(tail) if "vmcnt" in x)
    assert not any(re.match(r"(global|flat)_load_dword", x) for x in tail[:end])


def test_scan_bwd_folded_dma_walk_requests_stay_in_flight():
    ins = _body("scan_bwd.hip", "scan_bwd_kernelINS_6bf16_tELi8ELb1ELi16ELb1ELb1E")
    start = _after_last_request_group(ins, lambda x: "global_load_lds" in x, 5)
    assert _work_before_wait(ins, start, lambda x: x.startswith("v_")) > 600


def test_attention_dkv_tiles_stay_in_flight_under_the_mfmas():
    ins = _body("attn.hip", "attn_bwd_dkv_kernelINS_6bf16_tELi64ELi4ELb0ELb1E")
    start = _after_last_request_group(ins, lambda x: "global_load_lds" in x, 4)
    assert _work_before_wait(ins, start, lambda x: x.startswith("v_mfma")) >= 32


@pytest.mark.parametrize("name,mfmas", [("attn_q_kernelINS_6bf16_tELi128ELi4ELb1ELb0E", 16), ("attn_q_kernelINS_6bf16_tELi64ELi4ELb0ELb0E", 16),
                                        ("attn_bwd_dkv_kernelINS_6bf16_tELi128ELi4ELb0ELb0E", 16)])
def test_attention_register_staged_tiles_stay_in_flight_under_the_mfmas(name, mfmas):
    ins = _body("attn.hip", name)
    start = _after_last_request_group(ins, lambda x: x.startswith("global_load_dwordx4"), 4)
    assert _work_before_wait(ins, start, lambda x: x.startswith("v_mfma")) >= mfmas


def test_dir_merge_tile_loads_are_all_requested_before_the_first_wait():
    ins = _body("dir_perm.hip", "dir_merge_vec_kernelINS_6bf16_tE")
    first_wait = next(k for k, x in enumerate(ins) if "vmcnt" in x)
    assert sum(1 for x in ins[:first_wait] if x.startswith("global_load_dwordx4")) == 6
    assert sum(1 for x in ins[:first_wait] if x.startswith("global_load_")) >= 20


# ---- round 4: properties of the decode kernels and of the scan backward that no numerical test sees --------------------------------
def _scratch_ops(ins):
    return [x for x in ins if x.startswith("scratch_") or x.startswith("buffer_store_dword") and "offen" in x]


def test_scan_bwd_training_kernels_have_no_scratch():
    """the 8-wave dstate-16 walks (every training shape): the epilogue's lane / wave indices are re-derived, not kept alive across the
    chunk loop (they were the two spilled VGPRs of rounds 2-3)"""
    for name in ("scan_bwd_kernelINS_6bf16_tELi8ELb1ELi16ELb0ELb1E", "scan_bwd_kernelINS_6bf16_tELi8ELb1ELi16ELb1ELb1E",
                 "scan_bwd_kernelINS_5f16_tELi8ELb1ELi16ELb0ELb1E"):
        assert not _scratch_ops(_body("scan_bwd.hip", name)), name


@pytest.mark.parametrize("elt,suffix", [("EltBf16", "bf16"), ("EltF16", "f16")])
def test_beams_attention_runs_its_products_on_the_matrix_cores(elt, suffix):
    """decode_attn_beams_mfma_kernel: K Q^T by 16x16x32 MFMA, V^T P by 16x16x16 MFMA with V^T through the transpose read, tiles by
    LDS-DMA with explicit counted waits, no scratch; the only ds_bpermute left are the per-tile max over the four q-lanes and the
    final sum (the VALU version ran four dependent ones per (position, beam)).  Round 5: the same instruction stream for the fp16
    instantiation (csrc/decode_elt.h), which is the dtype the reference loads its LLM in."""
    ins = _body("decode.hip", f"decode_attn_beams_mfma_kernelINS_{len(elt)}{elt}ELi128ELi3ELi8E")
    count = lambda pat: sum(1 for x in ins if re.match(pat, x))
    assert count(rf"v_mfma_f32_16x16x32[_a-z0-9]*{suffix}") >= 4 and count(rf"v_mfma_f32_16x16x16[_a-z0-9]*{suffix}") >= 8
    assert count(r"ds_read_b64_tr_b16") >= 8 and count(r"global_load_lds_dwordx4") >= 8
    assert any(re.match(r"s_waitcnt vmcnt\(8\)", x) for x in ins), "the wait for the older of two tiles in flight"
    assert not _scratch_ops(ins)
    assert count(r"ds_bpermute_b32") <= 12      # 2 per tile (max over q) + 2 (final sum) + the fresh position's wave reduction


@pytest.mark.parametrize("elt", ["EltBf16", "EltF16"])
def test_per_row_decode_attention_sums_scores_by_dpp(elt):
    """decode_attn_kernel: the 16-lane score sum is four DPP adds; ds_bpermute only in the final merge of the lane groups (behind the
    position loop), and the cache rows arrive through the LDS ring"""
    ins = _body("decode.hip", f"decode_attn_kernelINS_{len(elt)}{elt}ELi128ELi8ELi4E")
    assert sum(1 for x in ins if x.startswith("v_add_f32_dpp")) >= 4
    assert sum(1 for x in ins if x.startswith("global_load_lds_dwordx4")) >= 4
    first_dma = next(k for k, x in enumerate(ins) if x.startswith("global_load_lds_dwordx4"))
    last_dma = max(k for k, x in enumerate(ins) if x.startswith("global_load_lds_dwordx4"))
    assert not any(x.startswith("ds_bpermute_b32") for x in ins[first_dma:last_dma]), "a cross-lane LDS round trip inside the position loop"


def test_attention_dq64_kernel_has_no_scratch():
    """attn_dq64_kernel (the dQ pass of the 4080-token block-causal attention): rounds 3-4 carried 12 bytes of scratch -- the epilogue's row
    indices and the ragged-tile branch's lane index, stored before the key loop and reloaded behind it; both are re-derived from the
    work-item id now."""
    for name in ("attn_dq64_kernelINS_6bf16_tEEE", "attn_dq64_kernelINS_5f16_tEEE"):
        assert not _scratch_ops(_body("attn.hip", name)), name


@pytest.mark.parametrize("elt", ["EltBf16", "EltF16"])
@pytest.mark.parametrize("mt,r,nw,pf", [(5, 1, 3, 8), (5, 2, 3, 6), (2, 2, 3, 8)])
def test_wide_decode_projection_ring_is_asked_for_each_chunk_once(elt, mt, r, nw, pf):
    """decode_gemm_wide_kernel (the 17..80-row decoder projections; these are the instantiations Llama-2-7B's qkv and gate + up take at
    80 and 18 rows): a ring of PF stages of OPS = 2 R + ceil(2 MT / NW) LDS-DMA requests.  The prologue asks for PF - 1 stages and the
    loop body for one -- PF x OPS request instructions in the whole kernel, none in the unrolled tail (the first deep-ring version
    re-requested a chunk per tail iteration: a tenth of a launch's requests at 64 chunks); the tail's waits shrink with the ring,
    (PF - 2 - t) x OPS for t = 0 .. PF - 2; the loop body plus the PF - 1 tail copies hold PF x 2 R MT MFMAs; no scratch."""
    ins = _body("decode_gemm.hip", f"decode_gemm_wide_kernelINS_{len(elt)}{elt}ELi{mt}ELi{r}ELi{nw}ELi{pf}E")
    ops = 2 * r + (2 * mt + nw - 1) // nw
    assert sum(1 for x in ins if x.startswith("global_load_lds_dwordx4")) == pf * ops
    waits = {int(re.search(r"vmcnt\((\d+)\)", x).group(1)) for x in ins if "vmcnt" in x}
    assert {(pf - 2 - t) * ops for t in range(pf - 1)} <= waits, sorted(waits)
    assert sum(1 for x in ins if x.startswith("v_mfma_f32_16x16x32")) == pf * 2 * r * mt
    assert not _scratch_ops(ins)
