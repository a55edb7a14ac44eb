"""Static check of the counted memory waits in the built library.

The span-staging kernels (r512_fast_kernel.hip, r512x_kernel.hip, r256_kernel.hip) issue their LDS-DMA span loads in
assembly, invisible to the compiler's own wait counting, and wait for a span with `s_waitcnt vmcnt(N) expcnt(6)` (the
second field never waits; it marks these instructions), N = the number of
store instructions the step issues behind those loads (memory operations complete in issue order).  If the compiler ever
emitted fewer stores on some path than N, the wait would return before the span has landed.  This script takes the gfx950
code objects out of libdsp_stft_hip.so, disassembles them with llvm-objdump and reports, for every kernel that has such
waits: the N of those waits, the number of store instructions in every store sequence (what one pass through a step can
issue; see analyse) and the number of LDS-DMA loads per load sequence.  The sets of the first two must be equal, the third
one number.  It also reports any instruction that depends on M0 apart from those loads: the newer load form writes M0 before
every load and does not restore it.

    python tools/asm_waits.py [library]        prints one line per kernel and the verdict; exit code 1 on a mismatch
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def elf_section(path, name):
    """Bytes of one section of a 64-bit little-endian ELF file."""
    d = open(path, "rb").read()
    assert d[:4] == b"\x7fELF" and d[4] == 2 and d[5] == 1, "not a 64-bit little-endian ELF file"
    shoff, = struct.unpack_from("<Q", d, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", d, 0x3A)
    sh = [struct.unpack_from("<IIQQQQIIQQ", d, shoff + i * shentsize) for i in range(shnum)]
    stroff = sh[shstrndx][4]
    for s in sh:
        n = d[stroff + s[0]:d.index(b"\0", stroff + s[0])].decode()
        if n == name:
            return d[s[4]:s[4] + s[5]]
    raise KeyError(name)


def code_objects(lib, arch="gfx950"):
    """The device code objects of `arch` in the library's .hip_fatbin section (one bundle per translation unit)."""
    fat = elf_section(lib, ".hip_fatbin")
    out, i = [], 0
    while True:
        j = fat.find(MAGIC, i)
        if j < 0:
            break
        n, = struct.unpack_from("<Q", fat, j + len(MAGIC))
        p = j + len(MAGIC) + 8
        for _ in range(n):
            off, size, tsz = struct.unpack_from("<QQQ", fat, p)
            triple = fat[p + 24:p + 24 + tsz].decode()
            p += 24 + tsz
            if arch in triple and size:
                out.append(fat[j + off:j + off + size])
        i = j + 1
    return out


def disassemble(co):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co)
        f.flush()
        return subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout


FUNC = re.compile(r"^([0-9a-f]+) <(\S+)>:$")
INSN = re.compile(r"^\s+(\S+)(?:\s+(.*?))?\s*//\s*([0-9A-Fa-f]+):")


def functions(text):
    cur, name = None, None
    for line in text.splitlines():
        m = FUNC.match(line)
        if m:
            if cur is not None:
                yield name, cur
            name, cur = m.group(2), []
            continue
        m = INSN.match(line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), (m.group(2) or "").strip()))
    if cur is not None:
        yield name, cur


def analyse(insns):
    """(counted waits N > 0, store instructions per store sequence, LDS-DMA loads per load sequence) of one kernel.
    A store sequence is what one pass through a step can issue: the stores met in address order until an unconditional or
    backward branch, an LDS-DMA load, a counted wait, or a conditional branch that jumps over further stores (the
    alternatives of an if / else are separate sequences; a branch that only skips arithmetic between two stores, such as
    the lane tests before the last store of a row block, does not end one).  The control-flow graph is not followed beyond
    that: the compiler joins the alternatives through blocks whose exits depend on flags, and paths through them that no
    execution takes would be counted."""
    base = insns[0][0]
    def target(addr, args):
        # the operand of a branch is a signed count of dwords from the next instruction
        m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", args)
        if m:
            return base + int(m.group(1), 16)
        m = re.fullmatch(r"(-?\d+|0x[0-9a-fA-F]+)", args.split()[0]) if args else None
        if not m:
            return None
        v = int(m.group(1), 0) & 0xFFFF
        return addr + 4 + 4 * (v - 0x10000 if v & 0x8000 else v)
    is_store = lambda op: op.startswith(("global_store", "buffer_store", "flat_store"))
    store_addrs = [a for a, op, _ in insns if is_store(op)]
    def stores_between(a, b):
        return sum(1 for x in store_addrs if a < x < b)
    waits, seqs, dmas = [], [], []
    st = dm = 0
    def close():
        nonlocal st
        if st:
            seqs.append(st)
        st = 0
    for addr, op, args in insns:
        if op.startswith("global_load_lds"):
            close()
            dm += 1
            continue
        if dm and (op.startswith(("s_cbranch", "s_branch", "s_waitcnt vm")) or is_store(op) or op == "s_endpgm"):
            dmas.append(dm)
            dm = 0
        if op == "s_waitcnt":
            m = re.fullmatch(r"vmcnt\((\d+)\) expcnt\(6\)", args)       # the counted waits carry expcnt(6) as their mark
            if m:
                close()
                if int(m.group(1)) > 0:
                    waits.append(int(m.group(1)))
        elif is_store(op):
            st += 1
        elif op == "s_branch" or op == "s_endpgm" or op == "s_setpc_b64":
            close()
        elif op.startswith("s_cbranch"):
            t = target(addr, args)
            if t is None or t <= addr or stores_between(addr, t):
                close()
    close()
    if dm:
        dmas.append(dm)
    return waits, seqs, dmas


IMPLICIT_M0 = ("s_movrel", "v_movrel", "s_sendmsg", "ds_gws", "v_interp", "s_set_gpr_idx", "ds_param_load", "ds_direct_load")


def m0_readers(insns):
    """Instructions of a kernel, other than its LDS-DMA loads, that depend on the contents of M0.  ring_glds16_at writes M0 and
    leaves it changed, on the assumption that the compiler keeps nothing there: holds as long as this list is empty.  (Copies of
    M0 into a scalar register -- the older load form saves and restores it -- do not depend on what it holds.)"""
    out = []
    for addr, op, args in insns:
        if op.startswith("global_load_lds"):
            continue
        if op.startswith(IMPLICIT_M0):
            out.append(f"{op} {args}")
            continue
        srcs = [a.strip() for a in args.split(",")[1:]]
        if "m0" in srcs and not (op == "s_mov_b32" and args.split(",")[0].strip().startswith("s")):
            out.append(f"{op} {args}")
    return out


def check(lib):
    """List of (kernel, waits, store sequences, LDS-DMA sequences, problem or None) for every kernel with counted waits."""
    rows = []
    for co in code_objects(lib):
        text = disassemble(co)
        for name, insns in functions(text):
            if not any(k in name for k in ("stft_r512_fast_kernel", "stft_r512x_kernel", "stft_r256_kernel")) or not insns:
                continue
            waits, seqs, dmas = analyse(insns)
            if not waits:
                continue
            problems = []
            if set(seqs) != set(waits):
                low = [q for q in set(seqs) if q < min(waits)]
                problems.append(f"store instructions per sequence {sorted(set(seqs))} != counted waits {sorted(set(waits))}" +
                                (f" (UNSAFE: a sequence of {low} stores is shorter than every wait)" if low else ""))
            # (the number of loads does not enter a wait -- what the wait counts are the stores behind them -- but a kernel
            # issues a fixed number per span: NV, or, in the ring-of-quarters form of the 2048-point kernel, NV for a unit's
            # first frame and NV / 4 or NV / 2 (hop 512 or 1024) for the others)
            d = sorted(set(dmas))
            if not (len(d) == 1 or ("stft_r512x_kernel" in name and all(v in (d[-1] // 4, d[-1] // 2, d[-1]) for v in d))):
                problems.append(f"LDS-DMA loads per sequence: {d} (one fixed number expected, or NV / 4, NV / 2 and NV in the ring form)")
            m0 = m0_readers(insns)
            if m0:
                problems.append(f"M0 is read outside the LDS-DMA loads: {m0[:3]}")
            rows.append((name, sorted(set(waits)), sorted(seqs), sorted(set(dmas)), "; ".join(problems) or None))
    return rows


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dsp_amd", "libdsp_stft_hip.so")
    rows = check(lib)
    bad = 0
    for name, waits, seqs, dmas, problem in rows:
        print(f"{name}: counted waits vmcnt{waits}, stores per sequence {seqs}, LDS-DMA loads per sequence {dmas}" + (f"  <-- {problem}" if problem else ""))
        bad += problem is not None
    print(f"{len(rows)} kernels with counted waits, {bad} mismatches")
    sys.exit(1 if bad or not rows else 0)
