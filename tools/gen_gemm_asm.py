#!/usr/bin/env python3
"""Generates april_asr_amd/csrc/gemm_mainloop_asm.inc: the hand-scheduled K loop of the 64x64 fp32 GEMM tile
(one wave: 4x4 accumulator tiles of v_mfma_f32_16x16x4_f32).

Why by hand: per k-block a wave issues 64 MFMAs (32 cycles of matrix pipe each) and 8 sixteen-byte loads plus
their address arithmetic.  Left to the compiler the loads and SALU/VALU work are emitted in one clump after the 64
MFMAs, where nothing hides them (83 % of the MFMA peak at one wave per SIMD, measured); attempts to interleave them
with sched_group_barrier or source order made the register allocator shuttle the 64 accumulators between AGPRs and
VGPRs every iteration.  Here every group of 8 MFMAs is followed by exactly one load (and a few scalar instructions),
which issue in the shadow of the matrix pipe.

Register plan (fixed registers, listed as clobbers; the accumulators and the lane offsets are compiler-allocated
operands): NB operand buffers of 32 VGPRs (4 weight quads, 4 activation quads) from v{BASE}; scalars s80..s88.
Block i computes from buffer i % NB while the loads of block i + NB - 1 land in the buffer block i - 1 released.
Loads past the last block re-read the last block (the scalar bases stop advancing), so nothing is read out of range.

usage: gen_gemm_asm.py > april_asr_amd/csrc/gemm_mainloop_asm.inc
"""
import sys

BASE = 112


def emit(nb, name, NT=4, scaled=False):
    out = []
    A = out.append

    BUF = NT * 4 + 16                      # registers per operand buffer: NT weight quads + 4 activation quads
    def bq(b, nt): return BASE + b * BUF + nt * 4
    def aq(b, mt): return BASE + b * BUF + NT * 4 + mt * 4
    def quad(r): return "v[%d:%d]" % (r, r + 3)

    def loads_and_advance(b, lines_between=None):
        """8 loads of the current fetch block into buffer b, then advance/park the bases; returns list of instruction strings"""
        ins = []
        for nt in range(NT):
            ins.append("global_load_dwordx4 %s, %%[boff%d], s[82:83]" % (quad(bq(b, nt)), nt))
        for mt in range(4):
            ins.append("global_load_dwordx4 %s, %%[aoff%d], s[80:81]" % (quad(aq(b, mt)), mt))
        adv = ["s_cmp_gt_u32 s84, 1", "s_cselect_b32 s86, 64, 0", "s_cselect_b32 s87, 1024, 0", "s_cselect_b32 s88, 1, 0",
               "s_sub_u32 s84, s84, s88", "s_add_u32 s80, s80, s86", "s_addc_u32 s81, s81, 0", "s_add_u32 s82, s82, s87", "s_addc_u32 s83, s83, 0"]
        return ins, adv

    A("s_mov_b64 s[80:81], %[ap]")
    A("s_mov_b64 s[82:83], %[bp]")
    A("s_mov_b32 s84, %[nblk]")
    A("s_mov_b32 s85, %[nblk]")
    for b in range(nb - 1):
        ld, adv = loads_and_advance(b)
        out.extend(ld); out.extend(adv)
    A("@ALIGN@")                         # APRIL_ASM_LOOP_ALIGN: a macro of kernels_gemm.hip (".p2align 3\n": the loop head on an 8-byte boundary)
    A("L_top_%=:")
    for b in range(nb):
        r = (b + nb - 1) % nb
        ld, adv = loads_and_advance(r)
        A("s_waitcnt vmcnt(%d)" % ((NT + 4) * (nb - 2)))
        muls = []
        if scaled:
            # AOP_SCALE: the activation quads of this block are multiplied in place by the lane's row scales (x = y * scale)
            # before the MFMAs that read them: the four k-step-0 components up front, the other twelve as fillers during
            # the k-step-0 MFMAs (k-step s is first read by MFMA 4*NT*s)
            for mt in range(4):
                A("v_mul_f32 v%d, v%d, %%[sc%d]" % (aq(b, mt), aq(b, mt), mt))
            A("s_nop 1")
            for ks in range(1, 4):
                for mt in range(4):
                    muls.append("v_mul_f32 v%d, v%d, %%[sc%d]" % (aq(b, mt) + ks, aq(b, mt) + ks, mt))
        mfma = []
        for ks in range(4):                   # k-step major, then m-tile, then n-tile: every accumulator gets its k-steps in order
            for mt in range(4):
                for nt in range(NT):
                    mfma.append("v_mfma_f32_16x16x4_f32 %%[c%d], v%d, v%d, %%[c%d]" % (mt * NT + nt, aq(b, mt) + ks, bq(b, nt) + ks, mt * NT + nt))
        # fillers ride in the shadow of the matrix pipe (the loads, then the scalar bookkeeping): one after every third
        # MFMA of the 64 of a 64x64 tile, one after (almost) every MFMA of the 32 of a 64x32 tile
        fill = muls + ld + adv + ["s_sub_u32 s85, s85, 1", "s_cmp_eq_u32 s85, 0"]
        every = (2 if scaled else 3) if NT == 4 else 1
        assert len(fill) * every + 1 <= len(mfma)
        placed = 0
        for i, m in enumerate(mfma):
            A(m)
            if i % every == every - 1 and fill:
                f = fill.pop(0)
                if f.startswith("v_mul_f32"):          # the multiply of k-step s must be issued before MFMA 4*NT*s
                    ks = 1 + placed // 4
                    assert i + 1 < 4 * NT * ks, (i, ks)
                    placed += 1
                A(f)
        assert not fill
        if b < nb - 1:
            A("s_cbranch_scc1 L_end_%=")
        else:
            A("s_cbranch_scc0 L_top_%=")
    A("L_end_%=:")
    A("s_waitcnt vmcnt(0)")
    A("s_nop 7"); A("s_nop 7"); A("s_nop 7")
    text = "".join(('    APRIL_ASM_LOOP_ALIGN\n' if l == "@ALIGN@" else '    "%s\\n"\n' % l) for l in out)
    clob = ", ".join('"v%d"' % r for r in range(BASE, BASE + nb * BUF))
    clob += ', "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "scc", "memory"'
    return "#define %s_TEXT \\\n%s\n#define %s_CLOBBERS %s\n" % (
        name, text.rstrip("\n").replace("\n", " \\\n"), name, clob)


if __name__ == "__main__":
    print("// GENERATED by tools/gen_gemm_asm.py -- do not edit.  See that file for the schedule and the register plan.")
    print(emit(2, "APRIL_MAINLOOP2"))
    print(emit(3, "APRIL_MAINLOOP3"))
    print(emit(2, "APRIL_MAINLOOP2_NT2", NT=2))
    # scaled=True variants (activation quads multiplied in place by a per-row scale, as fillers) were measured at +7 % loop
    # time on the gate GEMM and are not emitted: the row scale is applied to the finished partial sums instead
