#!/usr/bin/env python
"""Generate naf_amd/csrc/stem_rows_sched.inc: the hand-placed instruction schedule of the row-streaming 3x3 stem layer
(stem_conv_rows_kernel, stem_rows_kernel.h), as straight-line HIP.

    python tools/gen_stem_rows.py [out.inc]      # rewrites the .inc (committed; the build does not run this)

One wave per SIMD issues in order, so side work hides only in the 32-cycle shadow of individual MFMAs: every micro-op is
pinned behind the MFMA of its slot (sched_barrier + asm register anchors), a few independent instructions per slot.

The body is two double-steps (D = 0, 1) of 144 MFMA slots = 2 input rows x 24 fragments x 3 output rows.  Input row t
(t = 2 D + tt, names mod 4) feeds accumulator (t-1)&3 with tap row 2 (that output row is finished by it), t&3 with tap row 1
and (t+1)&3 with tap row 0 (that row starts here, from the conv bias); accumulator (t+2)&3 -- output row t-2 -- has its
epilogue during row t and is then re-initialised with the bias straight from the LDS.  Side work of a double-step: the scalar
row pointers / masks (u_all, slot 0), GroupNorm + SiLU + ring write of the five 16-byte pieces of batch d+2 (plain f32 VALU only,
ONE transcendental per micro-op), their reload with batch d+3 right after they are unpacked, the four row-store pieces of the
previous tile (tile read six slots ahead of its store), eight epilogue slices of three micro-ops each.  Placement: greedy, earliest
slot whose ISSUE-CYCLE budget (CAP cycles of the 32 an MFMA lasts) still has room, costs from profiles/r03_mfma_filler_prices.txt
(plain VALU 5.3, transcendental 9, LDS / scalar / VMEM a few), respecting the dependency gaps below.  The MFMA macro of the kernel
puts a sched_barrier behind every MFMA and NAF_SLOT_PIN one behind every slot, so what is placed here is what issues there.
"""
import os
import sys

NB = 3                      # B-fragment buffers
NLD, NST = 5, 4
NSLOT = 144
# Placement budget per MFMA gap, in issue cycles of one wave (profiles/r03_mfma_filler_prices.txt: a v_mfma_f32_32x32x16_bf16 issues
# every 32 cycles; beside it a plain VALU costs ~5.3 cycles of the wave's issue, a transcendental ~9, LDS / scalar / VMEM
# instructions a few): what does not fit the gap delays the next MFMA.
CAP = float(os.environ.get("NAF_ROWS_CAP", "25"))
V, TR, LR, LW, SA, VL, VS = 5.3, 9.0, 4.0, 5.0, 2.0, 8.0, 10.0   # plain VALU, transcendental, ds_read, ds_write_b64, SALU, load, store

def build(D):
    load = [0] * NSLOT                      # non-MFMA instructions already in the slot
    ops = [[] for _ in range(NSLOT)]        # (kind, code) behind the MFMA of the slot
    pre = [[] for _ in range(NSLOT)]        # before the MFMA of the slot

    # fragment requests: behind the last use (u == 2) of every fragment
    for k in range(NSLOT):
        if k % 3 == 2:
            load[k] += LR

    def place(cost, earliest, where=None):
        k = max(0, earliest)
        while k < NSLOT and load[k] + cost > CAP:
            k += 1
        assert k < NSLOT, "schedule does not fit"
        load[k] += cost
        return k

    # ---- the double-step's uniform row pointers and masks: first thing behind the first MFMA ----
    load[0] += 10 * SA
    ops[0].append(("uni", "u_all();"))

    # ---- epilogues: output row tt of the tile = accumulator (2 D + tt + 2) & 3, during input row tt ----
    for tt in range(2):
        nm = (2 * D + tt + 2) & 3
        k = 72 * tt + 3            # its last MFMA was slot 69 of the row before
        for j in range(4):
            # 12 plain instructions per slice, emitted as three micro-ops in consecutive slots
            for part in range(3):
                k = place((3 * V + LW) if part == 0 else 4 * V, k + 1)
                ops[k].append(("epi", f"epi{part}({nm}, {tt}, {j});"))
        # bias back into the accumulator: it is `sta` (tap row 0) of the next input row
        for j in range(4):
            k = place(LR, k + 1)
            ops[k].append(("epi", f"acc_init({nm}, {j});"))
        assert k < 72 * (tt + 1) - 4, "accumulator re-initialisation too late"

    # ---- row stores of the previous tile ----
    k = 40
    for n in range(NST):
        k = place(LR, k + 1)
        pre[k].append(("store", f"stv = *reinterpret_cast<const u32x4_t*>(prev_tile + st_lds0 + {16 * n} * PXE);"))
        k2 = place(VS + V, k + 6)     # the tile read has to come back first: LDS latency, in order behind the fragment reads
        ops[k2].append(("store", f"if (!EDGE || st_ok({n})) {{ uint32_t o_ = st_goff0; NAF_PIN1(o_); *reinterpret_cast<u32x4_t*>(prev_row{n // 2} + {'st_px16 + ' if n % 2 else ''}o_) = stv; }}"))
        k = k2

    # ---- commits (one variable set: piece n+1 starts after piece n has stored) ----
    start = 1
    for n in range(NLD):
        start = max(start, 1 + 27 * n)      # spread the pieces over the double-step
        cy, cu, co = "cy", "cu", "co"
        t = {}
        def put(name, cost, earliest, code, kind="commit"):
            k = place(cost, earliest)
            ops[k].append((kind, code))
            t[name] = k
            return k
        # Plain (non-packed) f32 VALU only: a v_pk_*_f32 beside an MFMA stalls the matrix pipe ~16 cycles, a plain v_fma does
        # not (profiles/r03_mfma_filler_prices.txt).  ys = log2(e) * GroupNorm(x) (the scale is folded into gav / gbv), so
        # SiLU(y) = ys * rcp(log2e + log2e * exp2(-ys)): the exp2's negation is an input modifier, the "1 +" an fma.
        for p in range(4):
            a_earliest = start if p == 0 else t[f"A{p - 1}"]
            put(f"A{p}", 2 * V, a_earliest,
                f"{{ const uint32_t w_ = ld[{n}][{p}]; cy[{2 * p}] = __uint_as_float(w_ << 16); cy[{2 * p + 1}] = __uint_as_float(w_ & 0xffff0000u); "
                f"NAF_PIN2(cy[{2 * p}], cy[{2 * p + 1}]); }}")
        # the piece's registers are free again: reload them with the same piece of batch d + 3
        r_lo, r_hi = (16 * n) // 40, (16 * n + 15) // 40
        ops[t["A0"]].append(("copy", f"NAF_LDS_WRITE_2X64(commit_base + c_off({n}), ld[{n}][0], ld[{n}][1], ld[{n}][2], ld[{n}][3]);"))
        if r_lo == r_hi:
            put("L", VL + V, t["A3"],
                f"{{ uint32_t o_ = col_off[{n}]; NAF_PIN1(o_); NAF_LD_DST({n}) = *reinterpret_cast<const u32x4_t*>(next_row{r_lo} + o_); }}", kind="load")
        else:   # the piece that straddles the batch's two rows: one base (the lower row), the row stride in the lane's offset
            put("L", VL + 2 * V, t["A3"],
                f"{{ uint32_t o_ = (straddle_hi != next_flip) ? col_off_s1 : col_off[{n}]; NAF_PIN1(o_); "
                f"NAF_LD_DST({n}) = *reinterpret_cast<const u32x4_t*>(next_lo + o_); }}", kind="load")
        for p in range(4):
            a, b = 2 * p, 2 * p + 1
            put(f"B{p}", 2 * V, t[f"A{p}"] + 1,
                f"{{ cy[{a}] = __builtin_fmaf(cy[{a}], gav[{a}], gbv[{a}]); cy[{b}] = __builtin_fmaf(cy[{b}], gav[{b}], gbv[{b}]); NAF_PIN2(cy[{a}], cy[{b}]); }}")
            for q, e in enumerate((a, b)):     # one transcendental per micro-op: two of them and anything else overfill a gap
                put(f"C{p}{q}", TR, t[f"B{p}"] + 1, f"{{ cu[{e}] = __builtin_amdgcn_exp2f(-cy[{e}]); NAF_PIN1(cu[{e}]); }}")
            put(f"D{p}", 2 * V, max(t[f"C{p}0"], t[f"C{p}1"]) + 2,
                f"{{ cu[{a}] = __builtin_fmaf(cu[{a}], kL, kL); cu[{b}] = __builtin_fmaf(cu[{b}], kL, kL); NAF_PIN2(cu[{a}], cu[{b}]); }}")
            for q, e in enumerate((a, b)):
                put(f"E{p}{q}", TR, t[f"D{p}"] + 1, f"{{ cu[{e}] = __builtin_amdgcn_rcpf(cu[{e}]); NAF_PIN1(cu[{e}]); }}")
            put(f"F{p}", 3 * V, max(t[f"E{p}0"], t[f"E{p}1"]) + 2,
                f"{{ const float r0_ = cy[{a}] * cu[{a}], r1_ = cy[{b}] * cu[{b}]; bf16x2_t o_; o_[0] = (bf16_t)r0_; o_[1] = (bf16_t)r1_; "
                f"co[{p}] = __builtin_bit_cast(uint32_t, o_); NAF_PIN1(co[{p}]); }}")
        g = put("G", 2 * LW + 2 * V, max(t[f"F{p}"] for p in range(4)) + 1,
                f"NAF_LDS_WRITE_2X64(commit_base + c_off({n}), co[0], co[1], co[2], co[3]);")
        start = g + 1
    last_commit = start

    # ---- POOL instantiation (naf_stem_conv_keys_fwd): the previous double-step's tile -> the cells' sums, see stem_rows_kernel.h ----
    # Per tile row g: the row's indicator operand, then four chains (fragment read -> small MFMA three slots later -> its result
    # joins the sums in the LDS two slots behind it; an asm MFMA has no hazard recogniser between it and its consumers).  In D = 0
    # the finished band's keys sit between the two rows, under `if (pfin)` (one double-step in eight).  Everything is placed on top of
    # the schedule above with a slightly larger budget (the 16-cycle MFMAs are not side work: they follow their slot's MFMA, and
    # the POOL instantiation has no GroupNorm sums in its epilogue slices); `if constexpr (POOL)` removes all of it elsewhere.
    PDIST = int(os.environ.get("NAF_ROWS_POOL_DIST", "3"))     # slots between a chain's LDS reads and its MFMA
    pool = [[] for _ in range(NSLOT)]
    PCAP = CAP + float(os.environ.get("NAF_ROWS_POOL_EXTRA", "5"))
    lp = list(load)
    for k in range(NSLOT):          # the epilogue slices' sums do not exist in this instantiation
        for kind, code in ops[k]:
            if code.startswith("epi1") or code.startswith("epi2"):
                lp[k] -= 4 * V
            elif code.startswith("epi0"):
                lp[k] -= V
    def pl(cost, earliest, cap=PCAP):
        k = max(0, earliest)
        while k < NSLOT - 1 and lp[k] + cost > cap:
            k += 1
        assert k < NSLOT - 1, "pool schedule does not fit"
        lp[k] += cost
        return k
    lb = list(lp)           # the keys' own budget (D = 0 only): one double-step in eight may run over
    BCAP = CAP + float(os.environ.get("NAF_ROWS_POOL_BND_EXTRA", "20"))
    def plb(cost, earliest):
        kk = max(0, earliest)
        while kk < NSLOT - 1 and lb[kk] + cost > BCAP:
            kk += 1
        assert kk < NSLOT - 1, "pool keys do not fit"
        lb[kk] += cost
        return kk
    k = 3                   # chain cursor: one accumulator register set, so the chains follow each other
    for g in range(2):
        ka = pl(LR + 2 * V, k)
        pool[ka].append(("", f"pool_a0(pr0 + {g});"))
        ka = pl(2 * V, ka + 2)
        pool[ka].append(("", "pool_a1();"))
        k = max(k, ka)
        for q in range(4):
            kr = pl(2 * LR, k)    # sums so far + tile fragment
            pool[kr].append(("", f"pool_rd(prev_tile, {g}, {q});"))
            km = pl(6.0, max(kr + PDIST, ka + 1))
            pool[km].append(("", "pool_mm();"))
            ks = pl(2 * LW, km + 2)
            pool[ks].append(("", f"pool_st({q}, 0); pool_st({q}, 1);"))
            k = ks
        if g == 0 and D == 0:       # the band's keys between the two rows' chains (not beside them: the chains' registers are free then)
            kf = k + 1
            for c in range(2):
                for i in range(4):
                    kf = plb(4 * LR + 2 * LW, kf)
                    pool[kf].append(("pfin", f"pool_f0(pr0, {c}, {i});"))
                    kf = plb(4 * V, kf + 3)
                    pool[kf].append(("pfin", f"pool_f1({i});"))
                    kf += 1
                for w in range(2):
                    kf = plb(4 * V, kf)
                    pool[kf].append(("pfin", f"pool_f2({w});"))
                    kf += 1
                kf = plb(4 * V + VS, kf)
                pool[kf].append(("pfin", f"pool_f3(pr0, {c});"))
                kf += 1
            k = kf
    pool_done = k

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
            elif kind == "copy":
                out.append(f"if constexpr (PLAIN && !(ABL & 1)) {{ {code} }}")
            else:
                out.append(f"if constexpr (!PLAIN && !(ABL & 1)) {{ {code} }}")
        for cond, code in pool[k]:
            if cond:
                out.append(f"if constexpr (POOL) {{ if ({cond}) {{ {code} }} }}")
            else:
                out.append(f"if constexpr (POOL) {{ {code} }}")
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
print(f"wrote {path}: 2 x {NSLOT} slots, cap {CAP} cycles; (max per slot, total side cycles, commits done by slot): {[(round(a, 1), round(b), c) for a, b, c in stats]}")
