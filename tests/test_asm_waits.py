"""The counted memory waits of the span-staging kernels, checked on the built library without a GPU.

tools/asm_waits.py takes the gfx950 code objects out of dsp_amd/libdsp_stft_hip.so, disassembles them and compares, for
every kernel that waits for its LDS-DMA span loads with a counted `s_waitcnt vmcnt(N)`, the N of those waits with the number
of store instructions the kernel's step can issue behind the loads (the two must agree: fewer stores than N and the wait
returns before the span has landed).  The same library built with -DDSP_SAFE_WAITS=1 must have no counted wait left; the
GPU suite compares the two builds bit for bit (tests/test_gpu_parity.py::test_counted_waits_against_full_waits)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_waits  # noqa: E402

LIB = os.path.join(ROOT, "dsp_amd", "libdsp_stft_hip.so")
SAFE = os.path.join(ROOT, "dsp_amd", "libdsp_stft_hip_safe.so")
need_objdump = pytest.mark.skipif(not os.path.exists(asm_waits.OBJDUMP), reason="llvm-objdump not in this image")


@need_objdump
def test_store_counts_match_the_counted_waits():
    rows = asm_waits.check(LIB)
    assert len(rows) >= 200, f"only {len(rows)} kernels with counted waits found: the mark or the extraction is broken"
    bad = [(name, problem) for name, _, _, _, problem in rows if problem]
    assert not bad, bad[:5]
    # the benchmark's kernel: 4 LDS-DMA loads per span, 5 sixteen-byte stores per step (17 four-byte ones when the block is unaligned)
    head = [r for r in rows if "stft_r512_fast_kernelIfLi13ELb0ELb0ELi0ELb0ELb1ELb0E" in r[0]]
    assert len(head) == 1 and head[0][1] == [5, 17] and head[0][2] == [5, 17, 17] and head[0][3] == [4], head


@need_objdump
def test_safe_build_has_no_counted_wait():
    assert os.path.exists(SAFE), "make -C dsp_amd/csrc builds libdsp_stft_hip_safe.so"
    assert asm_waits.check(SAFE) == []
    # ... but it does hold the kernels (their waits are all vmcnt(0) there)
    n = 0
    for co in asm_waits.code_objects(SAFE):
        for name, insns in asm_waits.functions(asm_waits.disassemble(co)):
            if "stft_r512_fast_kernel" in name or "stft_r512x_kernel" in name or "stft_r256_kernel" in name:
                n += any(op == "s_waitcnt" and args == "vmcnt(0) expcnt(6)" for _, op, args in insns)
    assert n >= 100


def test_checker_flags_a_short_store_sequence():
    """The analysis itself, on a made-up step: 4 LDS-DMA loads, a counted wait for 5, and one path with only 4 stores."""
    def seq(ops):
        return [(0x1000 + 4 * i, op, args) for i, (op, args) in enumerate(ops)]
    good = ([("global_load_lds_dwordx4", "v[0:1], off")] * 4 + [("v_add_f32", "v0, v1, v2")] + [("global_store_dwordx4", "v[0:1], v[2:5], off")] * 5 +
            [("s_waitcnt", "vmcnt(5) expcnt(6)"), ("s_cbranch_scc1", "65500"), ("s_endpgm", "")])
    waits, seqs, dmas = asm_waits.analyse(seq(good))
    assert waits == [5] and seqs == [5] and dmas == [4]
    bad = list(good)
    del bad[6]                                            # the compiler "merged" two stores
    waits, seqs, dmas = asm_waits.analyse(seq(bad))
    assert waits == [5] and seqs == [4]
    # an if / else with 5 and 17 stores, and a lane test between the stores of one block
    alt = ([("global_load_lds_dwordx4", "v[0:1], off")] * 4 + [("s_cbranch_vccnz", "8")] + [("global_store_dwordx4", "x")] * 5 + [("s_branch", "22")] +
           [("global_store_dword", "x")] * 16 + [("s_cbranch_execz", "2"), ("v_mov_b32", "v0, v1"), ("v_mov_b32", "v0, v1"), ("global_store_dword", "x")] +
           [("s_waitcnt", "vmcnt(17) expcnt(6)"), ("s_waitcnt", "vmcnt(5) expcnt(6)"), ("s_endpgm", "")])
    waits, seqs, dmas = asm_waits.analyse(seq(alt))
    assert sorted(waits) == [5, 17] and sorted(seqs) == [5, 17] and dmas == [4]
