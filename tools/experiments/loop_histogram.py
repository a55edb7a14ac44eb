"""Opcode histogram of the largest loop (backward branch) of a kernel of libdsp_stft_hip.so:
python tools/experiments/loop_histogram.py <substring of the mangled kernel name> [library]"""
import sys, os, re, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tools import asm_waits as A

pat = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "dsp_amd", "libdsp_stft_hip.so")
for co in A.code_objects(lib):
    text = A.disassemble(co)
    for name, insns in A.functions(text):
        if pat not in name:
            continue
        # insns: list of (addr, opcode, args)
        addr_index = {a: i for i, (a, _, _) in enumerate(insns)}
        best = None
        for i, (a, op, args) in enumerate(insns):
            if op.startswith("s_cbranch") or op == "s_branch":
                m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", args)
                if m:
                    tgt = insns[0][0] + int(m.group(1), 16)
                else:
                    try:
                        simm = int(args.split()[0], 0) & 0xFFFF
                    except Exception:
                        continue
                    if simm >= 0x8000:
                        simm -= 0x10000
                    tgt = a + 4 + 4 * simm
                if tgt < a and tgt in addr_index:
                    j = addr_index[tgt]
                    if best is None or i - j > best[1] - best[0]:
                        best = (j, i)
        if best is None:
            print(name, "no loop"); continue
        body = insns[best[0]:best[1] + 1]
        h = collections.Counter()
        for _, op, _ in body:
            key = op
            if op.startswith("v_pk_"): key = "v_pk_*"
            elif op.startswith("v_accvgpr"): key = "v_accvgpr_*"
            elif op.startswith("ds_read") or op.startswith("ds_load"): key = "ds_read*"
            elif op.startswith("ds_write") or op.startswith("ds_store"): key = "ds_write*"
            elif op.startswith("global_load_lds"): key = "global_load_lds"
            elif op.startswith("global_load"): key = "global_load*"
            elif op.startswith("global_store"): key = "global_store*"
            elif op.startswith("scratch_"): key = "scratch_*"
            elif op.startswith("s_waitcnt"): key = "s_waitcnt"
            elif op.startswith("s_nop"): key = "s_nop"
            elif op.startswith("v_"): key = "v_other:" + op.split("_e")[0]
            elif op.startswith("s_"): key = "s_other"
            h[key] += 1
        tot = len(body)
        valu = sum(c for k, c in h.items() if k.startswith("v_"))
        print(f"{name}\n  largest loop: {tot} instructions, vector {valu}, scalar {sum(c for k, c in h.items() if k.startswith('s_'))}")
        for k, c in h.most_common(40):
            print(f"    {c:5d} {k}")
