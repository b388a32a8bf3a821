"""tools/asm_hazard_check.py -- the build-time guard for inline-asm hazards hipcc does not pad.
It must flag the two bug patterns round 1 shipped and fixed (a196813: SGPR base fresh from v_readlane into an asm
global_load; c4ee55f: an asm VALU instruction reading an MFMA destination) and the wide-store pattern of the guide,
pass their fixed forms, and be green on the library as built."""
import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_hazard_check as H  # noqa: E402

PRE_A196813 = """
	v_readlane_b32 s4, v40, 0
	v_readlane_b32 s5, v40, 1
	;;#ASMSTART
	global_load_dwordx4 v[0:3], v10, s[4:5] sc1
	global_load_dwordx4 v[4:7], v11, s[4:5] sc1
	s_waitcnt vmcnt(0)
	;;#ASMEND
"""
POST_A196813 = PRE_A196813.replace(";;#ASMSTART\n", ";;#ASMSTART\n\ts_nop 4\n")

PRE_C4EE55F = """
	v_mfma_f32_16x16x4_f32 v[20:23], v1, v2, v[20:23]
	;;#ASMSTART
	v_min_f32_e64 v30, |v20|, v5
	;;#ASMEND
"""
# the fix made the first reads plain C: the compiler sees the dependency and pads it
POST_C4EE55F = """
	v_mfma_f32_16x16x4_f32 v[20:23], v1, v2, v[20:23]
	s_nop 11
	v_min_f32_e64 v31, |v20|, v5
	;;#ASMSTART
	v_fma_f32 v30, v31, v6, v7
	;;#ASMEND
"""

WIDE_STORE = """
	;;#ASMSTART
	s_nop 4
	global_store_dwordx4 v186, v[6:9], s[64:65]
	;;#ASMEND
	s_or_b64 exec, exec, s[22:23]
	v_and_b32_e32 v7, 64, v169
"""
WIDE_STORE_FIXED = WIDE_STORE.replace("s[64:65]\n", "s[64:65]\n\ts_nop 1\n")


def test_flags_sgpr_base_fresh_from_readlane():
    f = H.check_text(PRE_A196813)
    assert len(f) == 4 and all("[A]" in x for x in f), f     # two producers x two loads
    assert H.check_text(POST_A196813) == []


def test_flags_asm_valu_reading_mfma_destination():
    f = H.check_text(PRE_C4EE55F)
    assert len(f) == 1 and "[B]" in f[0], f
    assert H.check_text(POST_C4EE55F) == []


def test_accumulate_chain_is_not_a_hazard():
    chain = """
	v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]
	v_mfma_f32_16x16x32_bf16 v[0:3], v[16:19], v[20:23], v[0:3]
	;;#ASMSTART
	s_nop 0
	;;#ASMEND
"""
    assert H.check_text(chain) == []


def test_flags_wide_asm_store_whose_data_is_overwritten():
    f = H.check_text(WIDE_STORE)
    assert len(f) == 1 and "[C]" in f[0], f
    assert H.check_text(WIDE_STORE_FIXED) == []


def test_compiler_only_code_is_not_checked():
    """a dependency with neither end in an asm block is the compiler's business (its hazard recogniser pads it)"""
    text = """
	v_readlane_b32 s4, v40, 0
	v_readlane_b32 s5, v40, 1
	global_load_dwordx4 v[0:3], v10, s[4:5]
"""
    assert H.check_text(text) == []


def test_library_as_built_is_clean():
    paths = glob.glob(os.path.join(ROOT, "dlwp_benchmark_amd", "csrc", "build", "*-hip-amdgcn-amd-amdhsa-gfx950.s"))
    if not paths:
        pytest.skip("no device assembly beside the objects (GPU box: the build directory does not travel)")
    findings = []
    n_asm = 0
    for p in paths:
        f, _, a = H.check_file(p)
        findings += f
        n_asm += a
    assert n_asm > 1000, "the parser no longer sees the inline-asm blocks"
    assert findings == [], "\n".join(findings)
