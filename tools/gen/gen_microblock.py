#!/usr/bin/env python
"""Generates pyipm_amd/csrc/microblock_asm.inc: the 16 pivots of the in-register LDL' elimination of a 16x16
diagonal micro-block (tile_blocked.hpp), as four inline-asm statements of four pivots each.

Layout: lane i (i = lane & 15; the four 16-lane rows of a wave hold the same data) keeps row i of the micro-block
in a[0..15].  Pivot J (all lanes, no divergence):
    d     = a[J] of lane J                        v_mov_b64_dpp  row_newbcast:J   (the only cross-lane traffic:
    r     = 1/d    (rcp + two Newton steps)                       DP-ALU DPP, no LDS round trip, no readlane)
    u     = a[J] * clamp(i - J, 0, 1)             multiplier numerators of the rows below the pivot, 0 elsewhere
    a[J]  = -u r                                  = -l_iJ  (this is also M_i[J] of M = inv(L), see below)
    a[c] += a[c] of lane J * a[J]    c != J       v_fmac_f64_dpp row_newbcast:J
For c > J that is the Schur update of the not yet eliminated columns (the pivot ROW, by symmetry, is lane J's
registers); for c < J the same instruction applies the row operation to the augmented identity, so that after the
last pivot a[c], c < i, holds M = inv(L) (unit lower triangular, unit diagonal implicit).  One array, one formula.

Hazards handled by hand (the compiler does not look into inline asm): a VGPR written by a VALU instruction may be
read through DPP two wait states later at the earliest; the result of a transcendental (v_rcp_f64) needs one
independent instruction before its first VALU use on gfx940+.  The instruction ORDER below keeps both distances by
interleaving the previous pivot's remaining updates with the next pivot's dependent chain (software pipelining:
the chain mov_dpp -> rcp -> 4 fma -> mul -> first fmac is ~8 dependent DP instructions, the 14 other updates and the
bookkeeping fill its latency).

Two statements of eight pivots each (an asm statement takes at most 30 operands; the fewer statements, the fewer pivots
start with nothing to interleave: 4 x 4 pivots took 2180 cycles per micro-block, 2 x 8 take less).
Operands per statement: %0..%15 a[], %16..%23 d[8], %24 lmax, %25 ctr, %26 t, %27 u, %28 mk, %29 r (1/d of the pivot in
flight: dead once its multiplier column is out, so one register serves all pivots).
"""
import os

A = lambda c: "%%%d" % c
LMAX, CTR, T, U, MK, R = "%24", "%25", "%26", "%27", "%28", "%29"
NP = 8            # pivots per statement


def chain(J, b):
    """dependent chain of pivot J (block-local d/r operands), as a list of instructions"""
    d = "%%%d" % (16 + (J - NP * b))
    r = R
    return [
        "v_mov_b64_dpp %s, %s row_newbcast:%d row_mask:0xf bank_mask:0xf" % (d, A(J), J),
        "v_rcp_f64 %s, %s" % (r, d),
        # (two independent instructions are placed here by the interleaver: trans forwarding hazard)
        "v_fma_f64 %s, -%s, %s, 1.0" % (T, d, r),
        "v_fma_f64 %s, %s, %s, %s" % (r, r, T, r),
        "v_fma_f64 %s, -%s, %s, 1.0" % (T, d, r),
        "v_fma_f64 %s, %s, %s, %s" % (r, r, T, r),
        "v_mul_f64 %s, %s, -%s" % (A(J), U, r),                      # a[J] = -l
    ]


def pre(J):
    """off-chain preparation of pivot J: needs a[J] final (first update of pivot J-1)"""
    return [
        "v_add_f64 %s, %s, 0 clamp" % (MK, CTR),
        "v_add_f64 %s, %s, -1.0" % (CTR, CTR),
        "v_mul_f64 %s, %s, %s" % (U, A(J), MK),
    ]


def updates(J):
    order = list(range(J + 1, 16)) + list(range(0, J))
    return ["v_fmac_f64_dpp %s, %s, %s row_newbcast:%d row_mask:0xf bank_mask:0xf" % (A(c), A(c), A(J), J) for c in order]


def block(b):
    """NP pivots.  Order: for each pivot J: first update of J-1 is already out (it is issued right after the multiplier);
    then  pre(J), chain(J) with the REST of pivot J-1's updates interleaved between the chain's dependent instructions."""
    out = ["s_nop 1"]
    pending = []                                       # updates of the previous pivot not yet issued
    for J in range(NP * b, NP * b + NP):
        ch = chain(J, b)
        pr = pre(J)
        # the chain's instructions with fillers between them
        fill = pending + ["v_max_f64 %s, %s, |%s|" % (LMAX, LMAX, A(J - 1))] if J > NP * b else []
        seq = []
        seq += pr                                      # 3 instructions: also the DPP distance after the update that wrote a[J]
        k = 0
        nslots = len(ch) - 1
        per = (len(fill) + nslots - 1) // nslots if fill else 0
        for i, ins in enumerate(ch):
            seq.append(ins)
            if i < len(ch) - 1:
                take = fill[k:k + per]
                k += len(take)
                if i == 1 and len(take) < 2:           # after v_rcp: at least one independent instruction
                    take = take + ["s_nop 0"] * (1 - len(take)) if len(take) < 1 else take
                seq += take
        seq += fill[k:]
        out += seq
        ups = updates(J)
        out.append(ups[0])                             # column J+1 first: the next pivot's column
        pending = ups[1:]
    out += pending
    out.append("v_max_f64 %s, %s, |%s|" % (LMAX, LMAX, A(NP * b + NP - 1)))
    return out


def emit():
    lines = ["// GENERATED by tools/gen/gen_microblock.py -- do not edit.  See that file for what this is.",
             "// clang-format off"]
    for b in range(16 // NP):
        ins = block(b)
        lines.append("#define PYIPM_MICROBLOCK_ASM_%d(a, d, r, lmax, ctr, t, u, mk) \\" % b)
        lines.append("    asm volatile( \\")
        for s in ins:
            lines.append('        "%s\\n\\t" \\' % s)
        ops_out = ", ".join('"+v"(a[%d])' % c for c in range(16))
        ops_out += ", " + ", ".join('"=&v"(d[%d])' % (NP * b + k) for k in range(NP))
        ops_out += ', "+v"(lmax), "+v"(ctr), "=&v"(t), "=&v"(u), "=&v"(mk), "=&v"(r)'
        lines.append("        : %s \\" % ops_out)
        lines.append('        : : "memory")')
        lines.append("")
    lines.append("// clang-format on")
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    path = os.path.join(root, "pyipm_amd", "csrc", "microblock_asm.inc")
    with open(path, "w") as f:
        f.write(emit())
    print("wrote", path)
