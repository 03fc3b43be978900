#!/usr/bin/env python3
"""Checks the device assembly for uses of LDS read results that are still in flight.

    hipcc --offload-arch=gfx950 <flags of the Makefile> --cuda-device-only -S -o dev.s ycnr_als.hip
    python devtest/isa_lint.py dev.s [kernel-name-substring ...]

The Gramian kernels issue their LDS reads from inline asm and count the waits by hand
(`s_waitcnt lgkmcnt(N)` in later asm statements, MFMAs in between).  hipcc does not know that the
outputs of the reading asm are not valid yet: it may copy them to other registers, or use them,
before the waiting asm -- the copy then holds whatever the register held before the read for the
lanes whose data had not returned (LDS returns lanes 48..63 last).  This walks every kernel in
program order with a FIFO of outstanding LDS reads (LDS operations complete in order;
`s_waitcnt lgkmcnt(N)` retires all but the youngest N) and reports any instruction that reads or
writes a destination register of a read that has not been retired.  Compiler-generated reads must
come out clean too (the compiler waits before every use): that is the self-check of the tool.

Limits: straight-line walk (a loop back-edge is not followed; a branch does not reset the FIFO),
scalar memory loads share the counter but may return out of order and are ignored.

Second check (DPP): a DPP instruction must not read, through its DPP operand (src0), a VGPR that a
vector-ALU instruction wrote less than two wait states earlier (gfx9 data hazard "VALU writes VGPR ->
DPP reads that VGPR").  hipcc inserts the wait states between instructions it generates itself but not
around inline asm: the pivot updates of SolveMfmaF32 are inline-asm `v_fmac_f32_dpp` whose source is
produced by an asm multiply followed by `s_nop 1`, and a register copy (or an AGPR reload) the compiler
places directly in front of such a statement would break that silently.  The walk goes backwards from
every DPP instruction over two wait states (`s_nop N` counts N + 1), through labels into every branch
that targets them.  The same walk checks two single-wait-state rules: no v_readlane of a VGPR written by the
instruction directly in front of it, and no vector-ALU read of a transcendental result (v_rsq, ...) by the
instruction directly behind it -- an asm multiply placed right behind the compiler's v_rsq once read a stale
scale (wrong columns for k = 112 and 128, caught by tests/test_gpu_parity.py).
"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lint(path, wanted):
    text = open(path).read()
    total = 0
    for m in re.finditer(r"^(_Z[\w]+):[^\n]*$", text, re.M):
        name = m.group(1)
        if "ycnr" not in name or (wanted and not any(w in name for w in wanted)):
            continue
        end = text.find("s_endpgm", m.end())
        body = text[m.end():end].split("\n")
        # Only kernels that issue LDS reads from INLINE ASM have hand-counted waits to check.  In the others every
        # wait is the compiler's own, and the straight-line walk below (no back edges, no FIFO reset at branches)
        # reports false hazards across the basic blocks of their runtime loops (als_gen_solve_kernel).
        in_asm, asm_reads = False, 0
        for raw in body:
            if "#ASMSTART" in raw:
                in_asm = True
            elif "#ASMEND" in raw:
                in_asm = False
            elif in_asm and raw.split(";")[0].strip().startswith(("ds_read", "ds_bpermute")):
                asm_reads += 1
        if not asm_reads:
            continue
        fifo = []  # [(line number, instruction, dest regs)]
        reads = issues = 0
        for ln, raw in enumerate(body):
            ins = raw.split(";")[0].strip()
            if not ins or ins.endswith(":") or ins.startswith("."):
                continue
            op = ins.split()[0]
            if op == "s_branch":
                # nothing falls through an unconditional branch: what follows is reached by jumps only, whose state this
                # straight-line walk does not know (hipcc rotates loops, so the epilogue often sits behind the loop's last
                # block: without this the epilogue's first uses were charged to the reads of the loop body)
                fifo = []
                continue
            if op == "s_waitcnt":
                mm = LGKM.search(ins)
                if mm:
                    keep = int(mm.group(1))
                    fifo = fifo[max(0, len(fifo) - keep):] if keep else []
                elif "lgkmcnt" not in ins and "vmcnt" not in ins and "expcnt" not in ins:
                    fifo = []  # plain "s_waitcnt 0"
                continue
            pending = set().union(*[f[2] for f in fifo]) if fifo else set()
            touched = regs_of(ins.split(None, 1)[1]) if " " in ins else set()
            hit = pending & touched
            if hit:
                issues += 1
                src = next(f for f in fifo if f[2] & hit)
                if issues <= 5:
                    print(f"  {name[:70]}: '{ins}' touches v{sorted(hit)} of in-flight '{src[1]}' ({ln - src[0]} lines earlier)")
            if op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"):
                dest = regs_of(ins.split(None, 1)[1].split(",")[0])
                fifo.append((ln, ins, dest))
                reads += 1
            elif op.startswith("ds_"):
                fifo.append((ln, ins, set()))  # writes and others occupy a counter slot
        if reads:
            print(f"{'FAIL' if issues else 'ok  '} {name[:90]}: {reads} LDS reads, {issues} premature uses")
        total += issues
    return total


VMCNT = re.compile(r"vmcnt\((\d+)\)")


def lint_asm_vector_loads(path, wanted):
    """Vector-memory loads issued from INLINE ASM into registers (GramX6P fetches the ids and ratings of a step that way: a load
    the compiler counted would make it wait vmcnt(0) and drain the LDS-DMA ring) are retired by hand-counted `s_waitcnt vmcnt(N)`;
    until then nothing may read or copy their destination.  The walk keeps a FIFO of ALL vector-memory operations in program order
    (loads return in order; `vmcnt(N)` retires all but the youngest N), follows every loop back edge once more with the state it
    arrived with -- the value a loop carries is copied at the back edge, which is exactly where one build copied it one operation
    too early -- and reports any instruction that touches the destination of an in-flight asm load."""
    text = open(path).read()
    total = 0
    for m in re.finditer(r"^(_Z[\w]+):[^\n]*$", text, re.M):
        name = m.group(1)
        if "ycnr" not in name or (wanted and not any(w in name for w in wanted)):
            continue
        end = text.find("s_endpgm", m.end())
        raw_lines = text[m.end():end].split("\n")
        prog, labels, in_asm = [], {}, False  # (instruction, issued from inline asm)
        for raw in raw_lines:
            if "#ASMSTART" in raw:
                in_asm = True
                continue
            if "#ASMEND" in raw:
                in_asm = False
                continue
            ins = raw.split(";")[0].strip()
            if not ins or ins.startswith("."):
                if ins.endswith(":") and ins.startswith(".L"):
                    labels[ins[:-1]] = len(prog)
                continue
            if ins.endswith(":"):
                labels[ins[:-1]] = len(prog)
                continue
            prog.append((ins, in_asm))
        if not any(a and i.startswith(("buffer_load", "global_load")) and " lds" not in i for i, a in prog):
            continue
        issues, loads = 0, 0

        def is_vm(op):
            return op.startswith(("buffer_", "global_", "scratch_", "flat_"))

        def walk(start, stop, fifo, follow):
            nonlocal issues, loads
            i = start
            while i < stop:
                ins, from_asm = prog[i]
                op = ins.split()[0]
                if op == "s_branch":
                    tgt = ins.split()[-1]
                    if follow and tgt in labels and labels[tgt] <= i:  # an unconditional back edge
                        walk(labels[tgt], i + 1, list(fifo), False)
                    if not follow:
                        return
                    fifo[:] = []  # nothing falls through: what follows is reached by jumps only
                    i += 1
                    continue
                if op == "s_waitcnt":
                    mm = VMCNT.search(ins)
                    if mm:
                        keep = int(mm.group(1))
                        fifo[:] = fifo[max(0, len(fifo) - keep):] if keep else []
                    elif "lgkmcnt" not in ins and "expcnt" not in ins:
                        fifo[:] = []
                    i += 1
                    continue
                pending = set().union(*[f[1] for f in fifo]) if fifo else set()
                touched = regs_of(ins.split(None, 1)[1]) if " " in ins else set()
                hit = pending & touched
                if hit and not (is_vm(op) and from_asm and ins == [f[0] for f in fifo if f[1] & hit][0]):
                    issues += 1
                    if issues <= 5:
                        src = next(f for f in fifo if f[1] & hit)
                        print(f"  {name[:70]}: '{ins[:70]}' touches v{sorted(hit)} of in-flight '{src[0][:60]}'")
                if is_vm(op):
                    dest = set()
                    if from_asm and "load" in op and " lds" not in ins:
                        dest = regs_of(ins.split(None, 1)[1].split(",")[0])
                        loads += 1
                    fifo.append((ins, dest))
                if follow and op.startswith("s_cbranch"):
                    tgt = ins.split()[-1]
                    if tgt in labels and labels[tgt] <= i:  # a back edge: one more trip with the state of this one
                        walk(labels[tgt], i + 1, list(fifo), False)
                i += 1

        walk(0, len(prog), [], True)
        print(f"{'FAIL' if issues else 'ok  '} {name[:90]}: {loads} inline-asm vector loads, {issues} premature uses")
        total += issues
    return total


TRANS = ("v_rsq", "v_rcp", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")


def _dest_regs(ins):
    """VGPRs a vector-ALU instruction writes (first operand; both operands of a permlane swap)."""
    parts = ins.split(None, 1)
    if len(parts) < 2 or not parts[0].startswith("v_"):
        return set()
    ops = [o.strip() for o in parts[1].split(",")]
    if parts[0].startswith("v_permlane") and "swap" in parts[0]:
        return regs_of(ops[0]) | regs_of(ops[1])
    if parts[0].startswith("v_cmp") or parts[0].startswith("v_readlane") or parts[0].startswith("v_readfirstlane"):
        return set()
    return regs_of(ops[0])


def lint_dpp(path, wanted):
    text = open(path).read()
    total = checked = 0
    for m in re.finditer(r"^(_Z[\w]+):[^\n]*$", text, re.M):
        name = m.group(1)
        if "ycnr" not in name or (wanted and not any(w in name for w in wanted)):
            continue
        end = text.find("s_endpgm", m.end())
        ins_list, labels = [], {}
        for raw in text[m.end():end].split("\n"):
            ins = raw.split(";")[0].strip()
            if not ins or ins.startswith("."):
                if ins.endswith(":") and ins.startswith(".L"):
                    labels[ins[:-1]] = len(ins_list)
                continue
            if ins.endswith(":"):
                labels[ins[:-1]] = len(ins_list)
                continue
            ins_list.append(ins)
        sources = {}  # label position -> indices of the branches that jump there
        for i, ins in enumerate(ins_list):
            op = ins.split()[0]
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = ins.split()[-1]
                if tgt in labels:
                    sources.setdefault(labels[tgt], []).append(i)
        label_at = set(labels.values())

        def writers(i, slots, depth=0):
            """instructions in the `slots` wait states before instruction index i, over every path"""
            out = []
            j = i - 1
            while slots > 0:
                if j + 1 in label_at and depth < 4:
                    for b in sources.get(j + 1, []):
                        out += writers(b, slots, depth + 1)
                if j < 0:
                    break
                prev = ins_list[j]
                op = prev.split()[0]
                if op == "s_branch":  # no fall-through from here
                    break
                if op == "s_nop":
                    slots -= int(prev.split()[1]) + 1
                else:
                    out.append(prev)
                    slots -= 1
                j -= 1
            return out

        issues = 0
        for i, ins in enumerate(ins_list):
            op = ins.split()[0]
            if "_dpp" not in op:
                continue
            checked += 1
            ops = [o.strip() for o in ins.split(None, 1)[1].split(",")]
            src = regs_of(ops[1].split()[0]) if len(ops) > 1 else set()
            for w in writers(i, 2):
                if _dest_regs(w) & src:
                    issues += 1
                    if issues <= 5:
                        print(f"  {name[:70]}: '{ins[:60]}' reads v{sorted(src)} written by '{w}' less than two wait states earlier")
        # the two single-wait-state rules hipcc keeps for its own instructions but cannot keep for inline asm:
        # a v_readlane / v_readfirstlane of a VGPR the instruction before wrote, and a vector-ALU read of a
        # transcendental result (v_rsq, v_rcp, ...) by the instruction directly behind it
        for i, ins in enumerate(ins_list):
            parts = ins.split(None, 1)
            if len(parts) < 2 or not parts[0].startswith("v_"):
                continue
            ops = [o.strip() for o in parts[1].split(",")]
            lane_read = parts[0].startswith("v_readlane") or parts[0].startswith("v_readfirstlane")
            reads = regs_of(",".join(ops[1:]))
            if parts[0].startswith("v_fmac") or parts[0].startswith("v_mfma") or "swap" in parts[0]:
                reads |= regs_of(ops[0])
            for w in writers(i, 1):
                wop = w.split()[0]
                if wop == "v_writelane_b32":  # the compiler's own SGPR spill / reload pairs (no inline asm writes lanes): its hazard recogniser's business
                    continue
                hit = _dest_regs(w) & reads
                if hit and (lane_read or wop.startswith(TRANS)) and not parts[0].startswith(TRANS):
                    issues += 1
                    if issues <= 5:
                        print(f"  {name[:70]}: '{ins[:60]}' reads v{sorted(hit)} one instruction behind '{w[:50]}'")
        if issues:
            print(f"FAIL {name[:90]}: {issues} DPP reads too early")
        total += issues
    print(f"DPP instructions checked: {checked}")
    return total


def lint_scratch(path, limit=512):
    """Kernels whose registers spill to scratch memory (`; ScratchSize: N` in the assembly).  A change that
    looked harmless (inline-asm pivot updates) once made hipcc give up the accumulator half of the register
    file in one instantiation: 1104 bytes of scratch per lane and twice the run time, and no test noticed."""
    text = open(path).read()
    bad = 0
    for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+).*?\.amdhsa_private_segment_fixed_size (\d+)", text, re.M | re.S):
        name, size = m.group(1), int(m.group(2))
        if "ycnr" in name and size > limit:
            bad += 1
            print(f"FAIL {name[:90]}: {size} bytes of scratch per lane")
        elif "ycnr" in name and size > 64:
            print(f"note {name[:90]}: {size} bytes of scratch per lane")
    return bad


def lint_counted_waits_have_no_scratch(path):
    """A kernel that counts its vector-memory waits by hand -- inline-asm `s_waitcnt vmcnt(N)`, LDS-DMA (`buffer_load ... lds`) --
    must not carry ANY scratch: a spill's scratch_load / scratch_store is a vector-memory operation the count does not know,
    and where the compiler places it is not under the source's control.  Zero bytes proves every counted window clean."""
    text = open(path).read()
    sizes = {m.group(1): int(m.group(2)) for m in
             re.finditer(r"^\s*\.amdhsa_kernel (\S+).*?\.amdhsa_private_segment_fixed_size (\d+)", text, re.M | re.S)}
    bad = seen = 0
    for m in re.finditer(r"^(_Z[\w]+):[^\n]*$", text, re.M):
        name = m.group(1)
        if "ycnr" not in name:
            continue
        end = text.find("s_endpgm", m.end())
        body = text[m.end():end]
        counted = False
        in_asm = False
        for raw in body.split("\n"):
            if "#ASMSTART" in raw:
                in_asm = True
            elif "#ASMEND" in raw:
                in_asm = False
            elif in_asm and "vmcnt" in raw.split(";")[0]:
                counted = True
            elif re.match(r"\s*buffer_load_\w+ .*\blds\b", raw.split(";")[0]):
                counted = True
        if not counted:
            continue
        seen += 1
        n_scratch = len(re.findall(r"^\s*scratch_(load|store)", body, re.M))
        if sizes.get(name, 0) > 0 or n_scratch:
            bad += 1
            print(f"FAIL {name[:90]}: counts its vector-memory waits by hand and has {sizes.get(name, 0)} bytes of scratch "
                  f"({n_scratch} scratch instructions)")
    print(f"kernels with hand-counted vector-memory waits checked for scratch: {seen}")
    return bad


def lint_dual_occupancy(path):
    """Round 3-4: the dual class of 7 blocks solved 1 - 3 % of its rows wrong when it was built for two waves per SIMD.  Round 5
    (devtest/dual7/README.md): the wrong value is ONE right-hand-side block, and it is wrong exactly when hipcc has paired the updates
    b_bj -= U[J][bj]^T z_J of two blocks into v_pk_fma_f32 -- the shape of the stale-b hazard.  Without packed float32 instructions
    the two-wave build solves every row (lint_packed_fma below is the fence now).  What stays checked here: a dual class of 7 and
    more blocks that fits two waves per SIMD (<= 256 registers) must not contain a single packed float32 multiply."""
    text = open(path).read()
    bad = seen = 0
    for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.M | re.S):
        name, body = m.group(1), m.group(2)
        d = re.search(r"als_dual_solve_kernelILi(\d+)E", name)
        if not d or int(d.group(1)) < 7:
            continue
        nv = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        if not nv or int(nv.group(1)) > 256:
            continue
        seen += 1
        k = re.search(r"^" + re.escape(name) + r":[^\n]*$", text, re.M)
        end = text.find("s_endpgm", k.end()) if k else -1
        n = len(re.findall(r"^\s*v_pk_(?:fma|mul)_f32\b", text[k.end():end], re.M)) if k else 1
        if n:
            bad += 1
            print(f"FAIL {name[:90]}: dual class of {d.group(1)} blocks at two waves per SIMD with {n} packed float32 "
                  f"instructions: devtest/dual7/README.md")
    print(f"dual classes of 7+ blocks built for two waves per SIMD, checked for packed float32: {seen}")
    return bad


PACKED_F32 = re.compile(r"^\s*(v_pk_(?:fma|mul)_f32)\b", re.M)


def lint_packed_fma(path, kernels=None):
    """The fence of the stale-b and the dual7 hazard (devtest/pkfma/README.md): with the multiply-adds of two column blocks
    SLP-packed into v_pk_fma_f32 -- operand pairs assembled from kept copies, one multiplier broadcast by op_sel_hi -- kernels
    of this library computed wrong values under full occupancy (GramX6D: b in lanes 48..63; the 7-block dual class at two waves
    per SIMD: one block of the right-hand side), and right ones in every small test.  The library is built with
    -fno-slp-vectorize; a compiler that packs anyway would pass every CPU test and fail only on the GPU, so the build itself is
    checked: NO packed float32 multiply (v_pk_fma_f32 / v_pk_mul_f32) in ANY kernel.  (v_pk_add_f32 from source-level float2
    sums -- the LDS solver's column sums -- stays: the two-wave dual build with its 69 packed adds and without the paired
    multiply-adds solved every row.)"""
    text = open(path).read()
    bad = seen = 0
    for m in re.finditer(r"^(_Z[\w]+):[^\n]*$", text, re.M):
        name = m.group(1)
        if "ycnr" not in name or (kernels and not any(w in name for w in kernels)):
            continue
        seen += 1
        end = text.find("s_endpgm", m.end())
        found = PACKED_F32.findall(text[m.end():end])
        if found:
            bad += 1
            print(f"FAIL {name[:90]}: {len(found)} packed float32 instructions ({', '.join(sorted(set(found)))})")
    print(f"kernels checked for packed float32 arithmetic: {seen}")
    return bad


if __name__ == "__main__":
    n = lint(sys.argv[1], sys.argv[2:]) + lint_asm_vector_loads(sys.argv[1], sys.argv[2:])
    print("premature uses:", n)
    nd = lint_dpp(sys.argv[1], sys.argv[2:])
    print("DPP reads too early:", nd)
    ns = lint_scratch(sys.argv[1]) + lint_counted_waits_have_no_scratch(sys.argv[1]) + lint_dual_occupancy(sys.argv[1])
    print("kernels spilling to scratch / counted waits with scratch / two-wave dual classes with packed float32:", ns)
    npk = lint_packed_fma(sys.argv[1])
    print("kernels with packed float32 arithmetic:", npk)
    sys.exit(1 if n or nd or ns or npk else 0)
