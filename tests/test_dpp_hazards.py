"""Static ISA check (no GPU): no DPP operand of the built kernels is read within 2 wait states of a VALU write of the
same register (tools/check_dpp_hazards.py) -- the hazard pads that were dropped from the elimination row updates are
justified on the final code objects, and a future change of compiler or source that places a register copy in front of a
DPP group fails here instead of silently computing with a stale pivot row."""
import glob
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_dpp_hazards", os.path.join(ROOT, "tools", "check_dpp_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_checker_sees_a_hazard_and_its_fix():
    t = _tool()
    head = "0000000000001000 <k>:\n"

    def prog(lines):
        return head + "".join(f"\t{ln:60s}// {0x1000 + 4 * i:012X}: 00000000\n" for i, ln in enumerate(lines))
    dpp = "v_fmac_f32_dpp v1, v2, v3 row_newbcast:3 row_mask:0xf bank_mask:0xf"
    for lines, bad in (
            (["v_mov_b32_e32 v2, v9", dpp], 1),                                  # back to back
            (["v_mov_b32_e32 v2, v9", "v_add_f32_e32 v7, v7, v7", dpp], 1),      # one wait state
            (["v_mov_b32_e32 v2, v9", "s_nop 1", dpp], 0),                       # padded
            (["v_mov_b32_e32 v2, v9", "v_add_f32_e32 v7, v7, v7", "v_add_f32_e32 v8, v8, v8", dpp], 0),
            (["v_mov_b32_e32 v5, v9", dpp], 0),                                  # another register
            (["v_fma_f64 v[2:3], v[4:5], v[6:7], v[8:9]", "v_mov_b64_dpp v[10:11], v[2:3] row_newbcast:0 row_mask:0xf bank_mask:0xf"], 1),
            (["global_load_dword v2, v[4:5], off", dpp], 0),                     # not a VALU write (tracked by s_waitcnt)
            (["v_cmpx_lt_f32_e32 v4, v5", "s_nop 1", "s_nop 0", dpp], 1),          # VALU write of EXEC: 3 wait states, needs 5
            (["v_cmpx_lt_f32_e32 v4, v5", "s_nop 4", dpp], 0),
            # the writer sits in front of a branch whose target is the DPP instruction
            (["v_mov_b32_e32 v2, v9", "s_cbranch_scc1 2 <k+0x10>", "s_nop 1", "s_nop 1", dpp], 1),
    ):
        funcs = list(t.functions(prog(lines)))
        assert len(funcs) == 1
        assert len(t.check_function(funcs[0][1])) == bad, lines


def test_built_kernels_have_no_dpp_hazard():
    import __graft_entry__ as g
    g.build()
    t = _tool()
    objs = sorted(glob.glob(os.path.join(ROOT, "pyvbmp_amd", "csrc", "*.o")))
    assert objs, "no object files after build()"
    seen = 0
    for o in objs:
        report, ndpp, _ = t.check_object(o)
        seen += ndpp
        assert not report, "\n".join(report[:10])
    assert seen > 10000  # the elimination kernels really were inspected
