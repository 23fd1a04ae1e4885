#!/usr/bin/env python3
"""check_isa.py <cycle_kernel.s> [<cycle_feed_kernel.s>] -- build-time guard over the gfx950 assembly of the kernel TUs (run by the
Makefile right after the TUs are compiled and before either object exists; tests/test_capi_cpu.py runs it again and feeds it
deliberately broken builds).

The streaming kernels' keystream is one hand-scheduled assembly block per 16-byte word (cycle_kernel_impl.h,
ks_word_carry) that works in FIXED registers, v[120:127] and s[94:95], which the kernels keep out of the register
allocator's reach with amdgpu_num_vgpr(120) / amdgpu_num_sgpr(94).  Whether a compiler honours that is visible only in
its output, so the output is what is checked:
  * nothing outside the blocks touches the fixed registers, and no operand the compiler chose for a block lies in them
    (round 3: as plain clobbers the allocator handed them to inputs of the block -- wrong keystream, no error);
  * every block ends with the s_nop 0 that covers the SDWA dst_sel forwarding hazard towards the compiler's next instruction;
  * register counts stay inside the budget (the small shape: <= 64 VGPRs, 8 waves per SIMD), nothing spills, no scratch;
  * the work-queue kernel's ticket fetch is still ONE plain returning atomic per trip (LLVM's atomic optimizer would turn it
    into a wave-aggregated atomic followed at once by s_waitcnt vmcnt(0)), its mailbox is accessed with ds_ instructions,
    loads are nt, stores nt sc1, and the part table is read from the kernel arguments (no private segment).
For EVERY kernel of either TU, whatever its arithmetic (round 6):
  * no instruction directly behind an SDWA write with dst_sel BYTE_n / WORD_n reads the register that write touched (gfx940+:
    such a partial write needs one wait state before a VALU reads the register; LLVM's hazard recognizer does not look into
    inline assembly, and the small shape's put_byte is one such instruction per asm statement -- whether something else ended up
    between two of them was the instruction scheduler's habit, now it is a rule);
  * every s_barrier is reached with the EXEC mask the wave had when it entered the kernel: the kernel's control flow is
    followed with a stack of the masks saved by s_*_saveexec / narrowed by s_andn2 exec, and the stack must be EMPTY at a
    barrier on every path.  (The host-fed kernel's lab form hung a workgroup because lanes 1..63 of one wave went round the
    trip loop's back edge without lane 0 and met the barrier a second time; tools/archive/ubench_pcie_persist.hip.)
The host-fed kernel (cycle_feed_kernel.s) in particular: <= 64 VGPRs, no spills, no scratch, 8 bytes of LDS, exactly two
s_barrier, data loads nt, data stores sc1 and NOT nt (nt stores across PCIe measured 15-20 % slower), the trip's ticket and ok
word read into scalar registers (v_readfirstlane) behind the first barrier.
Exit status 0 = all of it holds; 1 = findings on stdout."""
import re
import sys

FIXED = re.compile(r"\bv12[0-7]\b|v\[\d+:12[0-7]\]|\bs9[45]\b|s\[\d+:9[45]\]")
OWN = re.compile(r"v\[12[0246]:12[1357]\]|s\[94:95\]|\bv12[0246]\b|\bv127\b")  # the block's own uses of them
BLOCK = re.compile(r";;#ASMSTART\n(.*?);;#ASMEND", re.S)


def kernel_bodies(asm):
    """mangled name -> text from its label to its s_endpgm"""
    out = {}
    for m in re.finditer(r"^(_Z\d+modgpu_cycle_\w+):", asm, re.M):
        out[m.group(1)] = asm[m.end():asm.index("s_endpgm", m.end())]
    return out


def kernel_texts(asm):
    """mangled name -> ALL of the kernel's text (a kernel may place blocks behind its first s_endpgm): up to its .Lfunc_end label"""
    out = {}
    for m in re.finditer(r"^(_Z\d+modgpu_cycle_\w+):", asm, re.M):
        end = re.compile(r"^\.Lfunc_end\d+:", re.M).search(asm, m.end())
        out[m.group(1)] = asm[m.end():end.start() if end else len(asm)]
    return out


def metadata(asm, name):
    """the scalar fields of one kernel's record in amdhsa.kernels (a record starts at "  - .agpr_count")"""
    meta = asm[asm.index("amdhsa.kernels"):]
    at = meta.index(".name:           " + name + "\n")
    start = meta.rfind("  - .agpr_count", 0, at)
    end = meta.find("  - .agpr_count", at)
    rec = meta[start:end if end > 0 else len(meta)]
    return {k: int(v) for k, v in re.findall(r"^\s+(?:- )?\.(\w+):\s+(\d+)\s*$", rec, re.M) if k not in ("offset", "size")}


# ---- rules for every kernel ------------------------------------------------------------------------------------------------------
INSN = re.compile(r"^\s+([a-z_0-9]+)\b(.*)$")


def instructions(fn):
    """[(label or None, mnemonic, operand text)] of a kernel body in text order; labels are attached to the instruction they precede"""
    out, pending = [], []
    for ln in fn.splitlines():
        code = ln.split(";")[0].rstrip()
        if not code.strip():
            continue
        m = re.match(r"^\s*(\.?[A-Za-z_][\w.$]*):", code)
        if m:
            pending.append(m.group(1))
            continue
        m = INSN.match(code)
        if not m or m.group(1).startswith("."):
            continue
        out.append((tuple(pending), m.group(1), m.group(2).strip()))
        pending = []
    return out


def vgprs(text):
    """the VGPR numbers an operand text names: v7, v[4:7]"""
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(x) for x in re.findall(r"\bv(\d+)\b", text))
    return regs


def sdwa_forwarding_hazards(name, fn):
    """gfx940+ hasDstSelForwardingHazard: one wait state between an SDWA write of part of a VGPR and a read of that VGPR"""
    bad = []
    ins = instructions(fn)
    for k, (_, op, args) in enumerate(ins[:-1]):
        m = re.search(r"dst_sel:(BYTE|WORD)_\d", args)
        if not op.endswith("_sdwa") or not m:
            continue
        dst = vgprs(args.split(",")[0])
        _, nop, nargs = ins[k + 1]
        if nop in ("s_nop", "s_waitcnt", "s_sleep") or nop.startswith("s_"):
            continue  # any instruction in between is the wait state
        operands = [a.strip() for a in nargs.split(",")]
        reads = set()
        for i, a in enumerate(operands):
            # operand 0 of a VALU / load is written, not read -- unless the instruction keeps the rest of it (SDWA UNUSED_PRESERVE)
            # or it is a store / atomic, whose operands are all read
            writes_op0 = i == 0 and not ("UNUSED_PRESERVE" in nargs) and not re.match(r"(buffer|global|flat|ds)_(store|write|atomic)", nop)
            if not writes_op0:
                reads |= vgprs(a)
        if dst & reads:
            bad.append("%s: %s reads v%d directly behind the SDWA partial write `%s %s` (dst_sel forwarding hazard: one wait state needed)"
                       % (name, nop, min(dst & reads), op, args.split(" dst_sel")[0]))
    return bad


def barriers_at_full_exec(name, fn):
    """follows the control flow with a stack of saved EXEC masks; every s_barrier must be reached with the stack empty"""
    ins = instructions(fn)
    at = {}
    for k, (labels, _, _) in enumerate(ins):
        for lb in labels:
            at[lb] = k
    bad, seen, work = [], set(), [(0, ())]
    while work:
        k, stack = work.pop()
        while k < len(ins) and (k, stack) not in seen:
            seen.add((k, stack))
            _, op, args = ins[k]
            ops = [a.strip() for a in args.split(",")]
            if re.match(r"s_\w+_saveexec_b64", op):
                if ops[0] not in stack:
                    stack = stack + (ops[0],)
            elif op == "s_andn2_b64" and ops[0] == "exec" and ops[1] == "exec":  # a loop's lanes leaving one by one
                if ops[2] not in stack:
                    stack = stack + (ops[2],)
            elif op == "s_or_b64" and ops[0] == "exec" and ops[1] == "exec":  # back to the mask saved in ops[2] (and out of everything nested inside)
                if ops[2] in stack:
                    stack = stack[:stack.index(ops[2])]
            elif op in ("s_mov_b64", "s_and_b64", "s_andn2_b64", "s_xor_b64", "s_or_b64") and ops[0] == "exec" and not stack:
                if not (op == "s_xor_b64" and len(ops) == 3):
                    bad.append("%s: EXEC is rewritten outside any saved-mask region: %s %s" % (name, op, args))
            elif op == "s_barrier" and stack:
                bad.append("%s: an s_barrier can be reached with part of the wave masked off (saved masks on the way: %s)" % (name, ", ".join(stack)))
            if op == "s_endpgm":
                break
            if op == "s_branch":
                k = at[ops[0]]
                continue
            if op.startswith("s_cbranch_"):
                work.append((at[ops[0]], stack))
            k += 1
    return sorted(set(bad))


def check_feed(asm, name, fn):
    """the host-fed kernel of cycle_feed_kernel.hip"""
    bad = []
    md = metadata(asm, name)
    if md.get("vgpr_count", 999) > 64:
        bad.append("%s: %d VGPRs -- more than 64, fewer than 8 waves per SIMD" % (name, md.get("vgpr_count", 999)))
    if md.get("group_segment_fixed_size", -1) != 8:
        bad.append("%s: LDS is %s bytes, expected the 8 of the ticket / ok mailbox" % (name, md.get("group_segment_fixed_size")))
    if fn.count("s_barrier") != 2:
        bad.append("%s: %d s_barrier, expected 2 (one behind thread 0's region, one at the end of the trip)" % (name, fn.count("s_barrier")))
    loads = [ln for ln in fn.splitlines() if "buffer_load_dwordx4" in ln]
    stores = [ln for ln in fn.splitlines() if "buffer_store_dwordx4" in ln]
    if not loads or not all(ln.split(";")[0].rstrip().endswith(" nt") for ln in loads):
        bad.append("%s: a data load is not nt" % name)
    if not stores or not all(ln.split(";")[0].rstrip().endswith(" sc1") and " nt" not in ln.split(";")[0] for ln in stores):
        bad.append("%s: a data store is not `sc1` without nt (nt stores across PCIe: -15..20 %%)" % name)
    ins = instructions(fn)
    first_barrier = next((k for k, (_, op, _) in enumerate(ins) if op == "s_barrier"), None)
    if first_barrier is not None:
        # text order is not execution order: the trip's first barrier is the one followed by the LDS read of the mailbox
        follows = [k for k, (_, op, _) in enumerate(ins) if op == "s_barrier" and any(o.startswith("ds_read") for _, o, _ in ins[k + 1:k + 3])]
        if len(follows) != 1 or sum(1 for _, o, _ in ins[follows[0] + 1:follows[0] + 8] if o == "v_readfirstlane_b32") < 2:
            bad.append("%s: the ticket and the ok word are not read into scalar registers right behind the trip's first barrier" % name)
    if fn.count("v_add_u32_sdwa") != 15:
        bad.append("%s: keystream instruction mix changed (%d v_add_u32_sdwa, expected 15: ALG 1)" % (name, fn.count("v_add_u32_sdwa")))
    return bad


def check(asm):
    """one TU's assembly: the rules for every kernel, then those of the TU it is (the streaming kernels' or the host-fed kernel's)"""
    bad = []
    bodies = kernel_bodies(asm)
    for name, fn in kernel_texts(asm).items():
        bad += sdwa_forwarding_hazards(name, fn)
        bad += barriers_at_full_exec(name, fn)
    feed = [n for n in bodies if "modgpu_cycle_feed_kernel" in n]
    if feed:
        if len(bodies) != 1:
            return bad + ["the host-fed kernel's TU holds %d kernels, expected 1" % len(bodies)]
        md = metadata(asm, feed[0])
        if md.get("vgpr_spill_count", 0) or md.get("sgpr_spill_count", 0) or md.get("private_segment_fixed_size", 0) or "scratch_" in bodies[feed[0]]:
            bad.append("%s: spills, scratch or a private segment: %s" % (feed[0], md))
        return bad + check_feed(asm, feed[0], bodies[feed[0]])
    queue = [n for n in bodies if "modgpu_cycle_queue_kernel" in n]
    if len(queue) != 1:
        return bad + ["expected exactly one work-queue kernel, found %d" % len(queue)]
    n_carry_kernels = 0
    for name, fn in bodies.items():
        md = metadata(asm, name)
        if md.get("vgpr_count", 999) > 128 or md.get("sgpr_count", 999) > 102:
            bad.append("%s: register counts beyond the budget: %s" % (name, md))
        if md.get("vgpr_spill_count", 0) or md.get("sgpr_spill_count", 0) or md.get("private_segment_fixed_size", 0):
            bad.append("%s: spills or a private segment: %s" % (name, md))
        if "scratch_" in fn:
            bad.append("%s: scratch instructions" % name)
        # the small shape (one word per lane, launch-latency-bound sizes and every launch across PCIe) lives on occupancy: it has
        # no pipeline of its own, so it must keep 8 waves per SIMD, i.e. at most 64 VGPRs (512 per SIMD lane / 8)
        if "modgpu_cycle_kernelILi1ELi256E" in name and md.get("vgpr_count", 999) > 64:
            bad.append("%s: the small shape needs %d VGPRs -- more than 64, fewer than 8 waves per SIMD" % (name, md.get("vgpr_count", 999)))
        blocks = BLOCK.findall(fn)
        carry = [b for b in blocks if "s[94:95]" in b]
        if not carry:
            continue  # (a kernel without the block may use any register)
        n_carry_kernels += 1
        outside = BLOCK.sub("", fn)
        for ln in outside.splitlines():
            if FIXED.search(ln) and not ln.strip().startswith(";"):
                bad.append("%s: a fixed temporary is touched OUTSIDE the keystream blocks: %s" % (name, ln.strip()))
        for b in carry:
            lines = [ln for ln in b.splitlines() if ln.strip()]
            for ln in lines:
                if FIXED.search(OWN.sub("", ln)):
                    bad.append("%s: the compiler gave a block operand a fixed temporary: %s" % (name, ln.strip()))
            if not lines or lines[-1].split(";")[0].strip() != "s_nop 0":
                bad.append("%s: a keystream block does not end with s_nop 0 (dst_sel forwarding hazard)" % name)
            if len([ln for ln in lines if "v_addc_co_u32_sdwa" in ln]) != 15 or len([ln for ln in lines if "v_mad_u64_u32" in ln]) != 30:
                bad.append("%s: a keystream block is not 30 mads + 15 addc" % name)
    if n_carry_kernels != 2:
        bad.append("expected the keystream block in exactly the two streaming kernels, found it in %d" % n_carry_kernels)
    q = bodies[queue[0]]
    if "v_mbcnt" in q:
        bad.append("queue kernel: the atomic optimizer rewrote the ticket atomic (build the TU with -mllvm -amdgpu-atomic-optimizer-strategy=None)")
    if q.count("global_atomic_add") != 4:  # one ticket fetch per unrolled trip (2) + a helper's first tickets + the exit count
        bad.append("queue kernel: %d global_atomic_add, expected 4" % q.count("global_atomic_add"))
    if "flat_" in q:
        bad.append("queue kernel: flat_ accesses (the LDS mailbox must be ds_ instructions)")
    if q.count("ds_write_b32") != 3 or q.count("ds_read_b32") != 3:
        bad.append("queue kernel: ticket mailbox traffic changed: %d ds_write_b32, %d ds_read_b32" % (q.count("ds_write_b32"), q.count("ds_read_b32")))
    loads = [ln for ln in q.splitlines() if "buffer_load_dwordx4" in ln]
    stores = [ln for ln in q.splitlines() if "buffer_store_dwordx4" in ln]
    if not loads or not all(ln.rstrip().endswith(" nt") for ln in loads):
        bad.append("queue kernel: a data load is not nt")
    if len([ln for ln in stores if ln.rstrip().endswith("nt sc1")]) < 8:  # 4 words x 2 unrolled trips (+ the cold peel loop)
        bad.append("queue kernel: fewer than 8 nt sc1 stores")
    if q.count("v_addc_co_u32_sdwa") != 9 * 15 or "v_add_u32_sdwa" in q:  # 4 words x 2 unrolled trips + the peeled first chunk
        bad.append("queue kernel: keystream instruction mix changed (%d addc)" % q.count("v_addc_co_u32_sdwa"))
    return bad


def main():
    bad, n_kernels = [], 0
    for path in sys.argv[1:]:
        asm = open(path).read()
        bad += check(asm)
        n_kernels += len(kernel_bodies(asm))
    for b in bad[:40]:
        print("check_isa:", b)
    if bad:
        print("check_isa: %d finding(s) -- the kernel TUs must not ship like this" % len(bad))
        return 1
    print("check_isa: ok (%d kernels)" % n_kernels)
    return 0


if __name__ == "__main__":
    sys.exit(main())
