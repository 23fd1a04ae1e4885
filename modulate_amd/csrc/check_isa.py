#!/usr/bin/env python3
"""check_isa.py <cycle_kernel.s> -- build-time guard over the gfx950 assembly of the kernel TU (run by the Makefile right
after the TU is compiled; tests/test_capi_cpu.py runs it again and feeds it a deliberately broken build).

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


def metadata(asm, name):
    """the scalar fields of one kernel's record in amdhsa.kernels (a record starts at "  - .agpr_count")"""
    meta = asm[asm.index("amdhsa.kernels"):]
    at = meta.index(".name:           " + name + "\n")
    start = meta.rfind("  - .agpr_count", 0, at)
    end = meta.find("  - .agpr_count", at)
    rec = meta[start:end if end > 0 else len(meta)]
    return {k: int(v) for k, v in re.findall(r"^\s+(?:- )?\.(\w+):\s+(\d+)\s*$", rec, re.M) if k not in ("offset", "size")}


def check(asm):
    bad = []
    bodies = kernel_bodies(asm)
    queue = [n for n in bodies if "modgpu_cycle_queue_kernel" in n]
    if len(queue) != 1:
        return ["expected exactly one work-queue kernel, found %d" % len(queue)]
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
    asm = open(sys.argv[1]).read()
    bad = check(asm)
    for b in bad[:40]:
        print("check_isa:", b)
    if bad:
        print("check_isa: %d finding(s) -- the kernel TU must not ship like this" % len(bad))
        return 1
    print("check_isa: ok (%d kernels)" % len(kernel_bodies(asm)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
