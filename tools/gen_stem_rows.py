#!/usr/bin/env python
"""Generate naf_amd/csrc/stem_rows_sched.inc: the hand-placed instruction schedule of the row-streaming 3x3 stem layer
(stem_conv_rows_kernel, stem_rows_kernel.h), as straight-line HIP.

    python tools/gen_stem_rows.py [out.inc]      # rewrites the .inc (committed; the build does not run this)

One wave per SIMD issues in order, so side work hides only in the 32-cycle shadow of individual MFMAs: every micro-op is
pinned behind the MFMA of its slot (sched_barrier + asm register anchors), a few independent instructions per slot.

The body is two double-steps (D = 0, 1) of 144 MFMA slots = 2 input rows x 24 fragments x 3 output rows.  Input row t
(t = 2 D + tt, names mod 4) feeds accumulator (t-1)&3 with tap row 2 (that output row is finished by it), t&3 with tap row 1
and (t+1)&3 with tap row 0 (that row starts here, from the conv bias); accumulator (t+2)&3 -- output row t-2 -- has its
epilogue during row t and is then re-initialised with the bias straight from the LDS.  Side work of a double-step: GroupNorm +
SiLU + ring write of the five 16-byte pieces of batch d+2, their reload with batch d+3 right after they are unpacked, the four
row-store pieces of the previous tile, eight epilogue slices.  Placement: greedy, earliest slot with room (CAP non-MFMA
instructions per slot), respecting the dependency gaps below.
"""
import os
import sys

NB = 3                      # B-fragment buffers
NLD, NST = 5, 4
NSLOT = 144
CAP = int(os.environ.get("NAF_ROWS_CAP", "4"))

def build(D):
    load = [0] * NSLOT                      # non-MFMA instructions already in the slot
    ops = [[] for _ in range(NSLOT)]        # (kind, code) behind the MFMA of the slot
    pre = [[] for _ in range(NSLOT)]        # before the MFMA of the slot

    # fragment requests: behind the last use (u == 2) of every fragment
    for k in range(NSLOT):
        if k % 3 == 2:
            load[k] += 1

    def place(cost, earliest, where=None):
        k = max(0, earliest)
        while k < NSLOT and load[k] + cost > CAP:
            k += 1
        assert k < NSLOT, "schedule does not fit"
        load[k] += cost
        return k

    # ---- epilogues: output row tt of the tile = accumulator (2 D + tt + 2) & 3, during input row tt ----
    for tt in range(2):
        nm = (2 * D + tt + 2) & 3
        k = 72 * tt + 3            # its last MFMA was slot 69 of the row before
        ku = place(2, k - 2)
        ops[ku].append(("uni", f"u_mask({tt});"))
        for j in range(4):
            k = place(4, k + 1)    # 8 instructions: two slots' worth; occupy this slot and the next
            load[min(k + 1, NSLOT - 1)] += 4 if k + 1 < NSLOT else 0
            ops[k].append(("epi", f"epi({nm}, {tt}, {j});"))
            k += 1
        # bias back into the accumulator: it is `sta` (tap row 0) of the next input row
        for j in range(4):
            k = place(1, k + 1)
            ops[k].append(("epi", f"acc_init({nm}, {j});"))
        assert k < 72 * (tt + 1) - 4, "accumulator re-initialisation too late"

    # ---- row stores of the previous tile ----
    k = 40
    for n in range(NST):
        k = place(1, k + 1)
        if n % 2 == 0:      # the row pointer of this tile row, just ahead of its first store
            ku = place(3, max(k - 3, 0))
            ops[ku].append(("uni", f"u_prev({n // 2});"))
            k = max(k, ku)
        pre[k].append(("store", f"stv = *reinterpret_cast<const u32x4_t*>(prev_tile + st_lds0 + {16 * n} * PXE);"))
        k2 = place(1, k + 2)
        ops[k2].append(("store", f"if (!EDGE || st_ok({n})) *reinterpret_cast<u32x4_t*>(prev_row{n // 2} + {'st_px16 + ' if n % 2 else ''}st_goff0) = stv;"))
        k = k2

    # ---- commits (one variable set: piece n+1 starts after piece n has stored) ----
    start = 1
    for n in range(NLD):
        start = max(start, 1 + 28 * n)      # spread the pieces over the double-step
        cy, cu, co = "cy", "cu", "co"
        t = {}
        def put(name, cost, earliest, code, kind="commit"):
            k = place(cost, earliest)
            ops[k].append((kind, code))
            t[name] = k
            return k
        if n in (0, 2):      # image-row pointers of batch d + 3: rows 0 / 1 are first needed by the reloads of pieces 0 / 2
            ku = place(3, max(start - 1, 0))
            ops[ku].append(("uni", f"u_next({n // 2});"))
        for p in range(4):
            a_earliest = start if p == 0 else t[f"A{p - 1}"]
            put(f"A{p}", 2, a_earliest,
                f"{{ const uint32_t w_ = ld[{n}][{p}]; {cy}[{p}] = f32x2_t{{__uint_as_float(w_ << 16), __uint_as_float(w_ & 0xffff0000u)}}; NAF_PIN1({cy}[{p}]); }}")
        # the piece's registers are free again: reload them with the same piece of batch d + 3
        r_lo, r_hi = (16 * n) // 40, (16 * n + 15) // 40
        base = f"next_row{r_lo}" if r_lo == r_hi else f"(pl + {16 * n} >= PXR ? next_row1 : next_row0)"
        put("L", 1, t["A3"], f"ld[{n}] = *reinterpret_cast<const u32x4_t*>({base} + col_off[{n}]);", kind="load")
        for p in range(4):
            put(f"B{p}", 2, t[f"A{p}"] + 1,
                f"{{ {cy}[{p}] = {cy}[{p}] * gav[{p}] + gbv[{p}]; {cu}[{p}] = {cy}[{p}] * c2; NAF_PIN2({cy}[{p}], {cu}[{p}]); }}")
            put(f"C{p}", 2, t[f"B{p}"] + 1,
                f"{{ {cu}[{p}] = f32x2_t{{__builtin_amdgcn_exp2f({cu}[{p}][0]), __builtin_amdgcn_exp2f({cu}[{p}][1])}}; NAF_PIN1({cu}[{p}]); }}")
            put(f"D{p}", 1, t[f"C{p}"] + 2, f"{{ {cu}[{p}] = {cu}[{p}] + dconst; NAF_PIN1({cu}[{p}]); }}")
            put(f"E{p}", 2, t[f"D{p}"] + 1,
                f"{{ {cu}[{p}] = f32x2_t{{__builtin_amdgcn_rcpf({cu}[{p}][0]), __builtin_amdgcn_rcpf({cu}[{p}][1])}}; NAF_PIN1({cu}[{p}]); }}")
            put(f"F{p}", 2, t[f"E{p}"] + 2,
                f"{{ const f32x2_t r_ = {cy}[{p}] * {cu}[{p}]; bf16x2_t o_; o_[0] = (bf16_t)r_[0]; o_[1] = (bf16_t)r_[1]; "
                f"{co}[{p}] = __builtin_bit_cast(uint32_t, o_); NAF_PIN1({co}[{p}]); }}")
        g = put("G", 1, max(t[f"F{p}"] for p in range(4)) + 1,
                f"*reinterpret_cast<u32x4_t*>(commit_base + c_off({n})) = u32x4_t{{{co}[0], {co}[1], {co}[2], {co}[3]}};")
        start = g + 1
    last_commit = start

    # ---- emit ----
    out = []
    for k in range(NSLOT):
        tt, r = divmod(k, 72)
        f, u = divmod(r, 3)
        t_row = 2 * D + tt
        nm = [(t_row - 1) & 3, t_row & 3, (t_row + 1) & 3][u]
        dy = 2 - u
        dx, ks = divmod(f, 8)
        fi = tt * 24 + f
        widx = (dy * 3 + dx) * 8 + ks
        if u == 0 and f == 0:
            out.append(f"// ---- input row {t_row} of the body: finishes accumulator {(t_row - 1) & 3}, starts {(t_row + 1) & 3}; epilogue of {(t_row + 2) & 3}")
        if pre[k]:
            out.append("__builtin_amdgcn_sched_barrier(0);")
            for kind, code in pre[k]:
                out.append(f"if constexpr (!(ABL & 8)) {{ {code} }}")
            out.append("__builtin_amdgcn_sched_barrier(0);")
        # explicit register classes: accumulators and B fragments in VGPRs, weights in AGPRs except the 8 fragments of tap
        # (0, 0) (288 weight registers > 256 AGPRs).  An asm MFMA is invisible to hipcc's hazard recogniser; the schedule keeps
        # every accumulator's first VALU read >= 18 MFMA slots behind its last MFMA, B fragments come from ds_read (waitcnt is
        # still tracked per register) and nothing but MFMAs writes the accumulators between acc_init and the epilogue.
        wcls = "v" if widx < 8 else "a"
        out.append(f"NAF_MFMA(acc[{nm}], wreg[{widx}], bb[{fi % NB}], \"{wcls}\");  // slot {k}")
        if u == 2:
            nf = fi + NB
            if nf < 24:
                base = f"cur + {2 * D} * ROWE"
            elif nf < 48:
                base = f"cur + {2 * D + 1} * ROWE"
            else:
                base = "cur + 2 * ROWE" if D == 0 else "oth"
            out.append(f"load_frag({base}, {nf % 24}, bb[{fi % NB}]);")
        for kind, code in ops[k]:
            if kind == "store":
                out.append(f"if constexpr (!(ABL & 8)) {{ {code} }}")
            elif kind in ("epi", "uni"):
                out.append(code)
            elif kind == "load":
                out.append(f"if constexpr (!(ABL & 16)) {{ {code} }}")
            else:
                out.append(f"if constexpr (!(ABL & 1)) {{ {code} }}")
        out.append("NAF_SLOT_PIN;")
    return out, load, last_commit


text = ["// GENERATED by tools/gen_stem_rows.py -- do not edit.  One double-step (two input rows, 144 MFMA slots) of",
        "// stem_conv_rows_kernel per value of NAF_ROWS_D, side work pinned behind individual MFMAs."]
stats = []
for D in (0, 1):
    body, load, last_commit = build(D)
    text.append(f"#if NAF_ROWS_D == {D}")
    text += body
    text.append("#endif")
    stats.append((max(load), sum(load), last_commit))

path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "naf_amd", "csrc", "stem_rows_sched.inc")
if len(sys.argv) > 1:
    path = sys.argv[1]
with open(path, "w") as fh:
    fh.write("\n".join(text) + "\n")
print(f"wrote {path}: 2 x {NSLOT} slots, cap {CAP}; (max per slot, total side instructions, commits done by slot): {stats}")
