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
