#!/usr/bin/env python
"""Generate naf_amd/csrc/stem_conv_sched3.inc: the hand-placed instruction schedule of one step of the
3x3 stem layer (stem_conv_kernel<3>), as straight-line HIP.

Why a generator: one wave per SIMD issues everything in order, so the ~300 VALU/LDS/global instructions
of side work per step (GroupNorm+SiLU of the next rows, row stores of the previous tile, loads two steps
ahead, epilogue of output row 0) only hide if they sit in the 32-cycle shadow of individual MFMAs, a few
independent instructions per slot, never a dependent chain.  hipcc does not do that by itself (it leaves
the side work in clumps and sinks unanchored arithmetic to its use), so the slot assignment is spelled
out here and every micro-op is anchored (asm volatile "+v") behind the MFMA of its slot.

    python tools/gen_stem_sched.py [out.inc]  # rewrites the .inc (committed; the build does not run this)

Geometry (must match StemGeom<3>): RS = 2 output rows, KH = 4 k-steps per B-fragment set, 24 sets =
(4 input rows x 3 tap columns x 2 k-halves); input rows 1, 2 feed both output rows (8 MFMAs / set), rows
0 and 3 feed one (4 MFMAs / set): 144 MFMA slots per step.

MFMA order (NAF_STEM_ORDER):
  rows  input rows 0, 1, 2, 3 in turn (default).  Rows 0 and 3 feed ONE output row each: 2 x 24 MFMAs that write the
        accumulator the MFMA before them wrote, with a fragment read / side work between them.
  pair  (round 3 experiment, rejected) no MFMA ever follows one on its own accumulator: row 1, then rows 0 and 3
        interleaved fragment by fragment, then row 2; both epilogues behind the last MFMA.  The hypothesis -- a dependent
        MFMA behind side work loses accumulator forwarding -- is false on gfx950: tools/mfma_chain_probe.hip measures
        32.3-32.7 cycles per MFMA for 1, 2 or 3 rotating accumulators with 0-6 fillers between them
        (profiles/r03_mfma_chain_probe.txt); the MFMA-only core of the step takes the same time in both orders, and the
        16 bias registers that stay live to the end push the step body into scratch (0.30 -> 0.44 ms).
"""
import os
import sys

KS, RS, KH, NROW = 3, 2, 4, 4
NLD, NST = 5, 4
TW, PXR = 32, 40
NSETS = NROW * KS * (8 // KH)
ORDER = os.environ.get("NAF_STEM_ORDER", "rows")

# ---- sets (fragment groups) in issue order, and the MFMA slots ------------------------------------
# seq[p] = (input row i, tap column dx, k half kh); slot = (p, ks, g, dy, dx, kh)
if ORDER == "rows":
    seq = [(i, dx, kh) for i in range(NROW) for dx in range(KS) for kh in range(8 // KH)]
    blocks = [[p] for p in range(NSETS)]                      # every set alone
else:
    r1 = [(1, dx, kh) for dx in range(KS) for kh in range(8 // KH)]
    r2 = [(2, dx, kh) for dx in range(KS) for kh in range(8 // KH)]
    pq = []
    for dx in range(KS):
        for kh in range(8 // KH):
            pq += [(0, dx, kh), (3, dx, kh)]
    seq = r1 + pq + r2
    blocks = [[p] for p in range(6)] + [[6 + 2 * q, 7 + 2 * q] for q in range(6)] + [[p] for p in range(18, 24)]
assert len(seq) == NSETS and sorted(seq) == sorted((i, dx, kh) for i in range(NROW) for dx in range(KS) for kh in range(2))

slots = []          # (p, ks, g, dy, dx, kh)
set_first_slot = {}
for blk in blocks:
    for ks in range(KH):
        for p in blk:
            i, dx, kh = seq[p]
            for g in range(RS):
                dy = i - g
                if 0 <= dy < KS:
                    set_first_slot.setdefault(p, len(slots))
                    slots.append((p, ks, g, dy, dx, kh))
NSLOT = len(slots)
assert NSLOT == 144
for a, b in zip(slots, slots[1:]):
    if ORDER != "rows":
        assert a[2] != b[2], "two MFMAs in a row on one accumulator"
ops = [[] for _ in range(NSLOT)]

# ---- micro-ops ------------------------------------------------------------------------------------
def commit_ops(n, v):
    """GroupNorm affine + SiLU + bf16 of load piece n (8 channels of one ring pixel), variable set v.
    y = x*ga + gb ; u = x*ga2 + gb2 = -log2(e)*y ; out = y * rcp(1 + exp2(u))."""
    cy, cu, co = f"cy{v}", f"cu{v}", f"co{v}"      # f32x2_t [4] pairs (v_pk_fma / v_pk_add / v_pk_mul)
    o = {}
    for p in range(4):
        o[f"A{p}"] = (f"{{ const uint32_t w_ = ld[{n}][{p}]; {cy}[{p}] = f32x2_t{{__uint_as_float(w_ << 16), "
                      f"__uint_as_float(w_ & 0xffff0000u)}}; NAF_PIN1({cy}[{p}]); }}")
        o[f"B{p}"] = (f"{{ const f32x2_t x_ = {cy}[{p}]; {cy}[{p}] = x_ * gav[{p}] + gbv[{p}]; {cu}[{p}] = x_ * ga2v[{p}] + gb2v[{p}]; "
                      f"NAF_PIN2({cy}[{p}], {cu}[{p}]); }}")
        o[f"C{p}"] = (f"{{ {cu}[{p}] = f32x2_t{{__builtin_amdgcn_exp2f({cu}[{p}][0]), __builtin_amdgcn_exp2f({cu}[{p}][1])}}; "
                      f"NAF_PIN1({cu}[{p}]); }}")
        o[f"D{p}"] = f"{{ {cu}[{p}] = {cu}[{p}] + 1.0f; NAF_PIN1({cu}[{p}]); }}"
        o[f"E{p}"] = (f"{{ {cu}[{p}] = f32x2_t{{__builtin_amdgcn_rcpf({cu}[{p}][0]), __builtin_amdgcn_rcpf({cu}[{p}][1])}}; "
                      f"NAF_PIN1({cu}[{p}]); }}")
        o[f"F{p}"] = (f"{{ const f32x2_t r_ = {cy}[{p}] * {cu}[{p}]; bf16x2_t o_; o_[0] = (bf16_t)r_[0]; o_[1] = (bf16_t)r_[1]; "
                      f"{co}[{p}] = __builtin_bit_cast(uint32_t, o_); NAF_PIN1({co}[{p}]); }}")
    o["G"] = (f"*reinterpret_cast<u32x4_t*>(ring + commit_slot * ROWE + c_off[{n}]) = "
              f"u32x4_t{{{co}[0], {co}[1], {co}[2], {co}[3]}};")
    return o

PLANS = {
    # 12 rows, pieces pipelined every 11 slots (several rows carry a transcendental pair plus other work)
    "dense": (11, [["A0", "A1"], ["B0", "B1"], ["C0", "A2"], ["C1", "D0", "A3"], ["E0", "D1", "B2"], ["E1", "B3"],
                   ["C2", "F0"], ["C3", "F1", "D2"], ["E2", "D3"], ["E3", "F2"], ["F3"], ["G"]]),
    # a transcendental pair (2 x 16 cycles) has its MFMA slot to itself
    "lone": (14, [["A0", "A1"], ["B0", "B1"], ["C0"], ["C1"], ["D0", "D1", "A2", "A3"], ["E0"], ["E1"], ["B2", "B3", "F0"],
                  ["C2"], ["C3"], ["F1", "D2", "D3"], ["E2"], ["E3"], ["F2", "F3"], ["G"]]),
}
PLAN = os.environ.get("NAF_STEM_PLAN", "dense")
COMMIT_STRIDE, PIECE_PLAN = PLANS[PLAN]
COMMIT_BASE = 2
for n in range(NLD):
    o = commit_ops(n, n & 1)
    for r, names in enumerate(PIECE_PLAN):
        for nm in names:
            ops[COMMIT_BASE + COMMIT_STRIDE * n + r].append(("commit", o[nm]))
last_a = max(r for r, names in enumerate(PIECE_PLAN) if any(x.startswith("A") for x in names))
last_commit_read = COMMIT_BASE + COMMIT_STRIDE * (NLD - 1) + last_a      # last A-stage of the last piece
last_commit = COMMIT_BASE + COMMIT_STRIDE * (NLD - 1) + len(PIECE_PLAN) - 1

# Row stores of the previous output tile and the global loads for the step after next.  Loads and stores share vmcnt and
# retire in order; either order measures the same (the stores have long reached L2 when the next step's GroupNorm micro-ops
# ask for their loads).
FIRST_BASE = max(60, last_commit + 1)      # ld[n] was consumed by A-stages before this slot
assert FIRST_BASE > last_commit_read
# The LDS answers in order, so a wait for ONE value is a wait for every read issued before it.  A row store's tile read
# therefore goes BEFORE the B-fragment read of its slot: its wait then leaves the fragment reads in flight.
pre = [[] for _ in range(NSLOT)]            # micro-ops emitted before the MFMA of a slot
if ORDER == "rows":
    cand = sorted(set_first_slot.values())
else:
    cand = list(range(0, NSLOT, 4))
st_slots = [k for k in cand if k >= FIRST_BASE][:NST]
assert len(st_slots) == NST and st_slots[-1] + 2 < 112
for n, k in enumerate(st_slots):
    pre[k].append(("store", f"stv = *reinterpret_cast<const u32x4_t*>(prev_tile + st_lds[{n}]);"))
    ops[k + 2].append(("store", f"if (!EDGE || st_ok(pst, {n})) *reinterpret_cast<u32x4_t*>(prev_row{16 * n // TW} + st_goff[{n}]) = stv;"))
ld_slots = [k for k in range(st_slots[0] + 3, 118) if k not in st_slots and (k - 2) not in st_slots and k % 2 == 1][:NLD]
assert len(ld_slots) == NLD
for n, k in enumerate(ld_slots):
    r_lo, r_hi = (16 * n) // PXR, (16 * n + 15) // PXR
    base = f"next_row{r_lo}" if r_lo == r_hi else f"(pl + {16 * n} >= {PXR} ? next_row1 : next_row0)"
    ops[k].append(("load", f"ld[{n}] = *reinterpret_cast<const u32x4_t*>({base} + col_off[{n}]);"))

# Epilogues.  An epilogue that reads its bias from the LDS on the spot waits for every LDS read issued before it (the LDS
# answers in order): ~100+ idle MFMA cycles, 8 x a step.
tail = []
if ORDER == "rows":
    # Row 0's accumulator is final once input row 2 is done (slot 119): its epilogue hides behind row 3's MFMAs, the lane's 16
    # bias values requested a few slots ahead (the GroupNorm temporaries are dead by then).  Row 1, whose epilogue is the exposed
    # tail of the step: the accumulator STARTS as the bias (four ds_read_b128 straight into its 16 registers at the top of the
    # step, before its first MFMA at slot 25), so its epilogue adds nothing.
    first_row3 = next(k for k, s in enumerate(slots) if seq[s[0]][0] == 3)
    assert first_row3 == 120
    first_g1 = next(k for k, s in enumerate(slots) if s[2] == 1)
    assert first_g1 == 25
    for j in range(4):
        ops[first_row3 - 10 + 2 * j].append(("epi", f"bj[{j}] = *reinterpret_cast<const f32x4_t*>(cvec + wave * 32 + 8 * {j} + 4 * half);"))
        ops[first_row3 + 2 + 5 * j].append(("epi", f"epi(0, {j}, T{{}});"))
        ops[4 + 5 * j].append(("epi", f"acc_init(1, {j});"))
    tail = [f"epi(1, {j}, F{{}});" for j in range(4)]
    zero_first = {0}          # output rows whose first MFMA starts from 0
else:
    # Both accumulators start from 0 and finish with the last two MFMAs: bias values requested a few slots before the end, both
    # epilogues behind the last MFMA (row 0's first: its last MFMA is the older one).
    for j in range(4):
        ops[NSLOT - 14 + 2 * j].append(("epi", f"bj[{j}] = *reinterpret_cast<const f32x4_t*>(cvec + wave * 32 + 8 * {j} + 4 * half);"))
    tail = [f"epi({g}, {j}, T{{}});" for g in range(RS) for j in range(4)]
    zero_first = {0, 1}

# B fragments: a rolling window of 8 register quads, bb[p & 1][ks] for set p.  The register of fragment (p, ks) is re-requested
# with fragment (p + 2, ks) right behind the last MFMA that reads it, so every fragment is in flight for 7+ MFMA slots (224+
# cycles).  Sets 24, 25 are sets 0, 1 of the NEXT step; CARRY of them are requested before the barrier (their input row must
# have been in the ring for a whole step: row 0 in `rows` order, row 1 in `pair` order -- the next step's row 1 is this step's
# row 3), the others at the top of the step.
last_slot_of = {}
for k, sl in enumerate(slots):
    last_slot_of[(sl[0], sl[1])] = k
CARRY = int(os.environ.get('NAF_STEM_CARRY', '1' if ORDER == "rows" else '2'))   # must match NAF_STEM_CARRY of the header
if ORDER == "rows":
    assert CARRY <= 2 and all(seq[c][0] == 0 for c in range(CARRY))
else:
    assert CARRY <= 2 and all(seq[c][0] == 1 for c in range(CARRY))

def frag(p, ks):
    nxt = 1 if p >= NSETS else 0
    i, dx, kh = seq[p - nxt * NSETS]
    return f"load_frag({nxt}, {i}, {dx}, {kh}, {ks}, bb[{p & 1}][{ks}]);"

# ---- emit -------------------------------------------------------------------------------------------
out = []
out.append("// GENERATED by tools/gen_stem_sched.py -- do not edit.  One step of stem_conv_kernel<3>: 144 MFMA slots,")
out.append(f"// side work pinned behind individual MFMAs (MFMA order: {ORDER}).  Included inside stem_conv_kernel.h: once with")
out.append("// NAF_SCHED_PROLOGUE (the fragment sets step 0 is entered with), once inside step_body.")
out.append("#ifdef NAF_SCHED_PROLOGUE")
out.append(f"static_assert(NAF_STEM_CARRY == {CARRY}, \"header and schedule disagree on the carried fragment sets\");")
for c in range(CARRY):
    for f in range(KH):
        i, dx, kh = seq[c]
        out.append(f"load_frag0({i}, {dx}, {kh}, {f}, bb[{c}][{f}]);")
out.append("#else")
for c in range(CARRY, 2):
    for f in range(KH):
        out.append(frag(c, f))
started = set()
for k, (p, ks, g, dy, dx, kh) in enumerate(slots):
    if set_first_slot[p] == k:
        i = seq[p][0]
        out.append(f"// ---- set {p}: input row {i}, tap column {dx}, k-steps {kh * KH}..{kh * KH + KH - 1}")
    if pre[k] or set_first_slot[p] == k:
        out.append("__builtin_amdgcn_sched_barrier(0);")
        for kind, code in pre[k]:
            out.append(f"if constexpr (!(ABL & 8)) {{ {code} }}")
        out.append("__builtin_amdgcn_sched_barrier(0);")
    widx = (dy * KS + dx) * 8 + kh * KH + ks
    src = f"acc[{g}]"
    if g in zero_first and g not in started:
        src = "f32x16_t{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}"
    started.add(g)
    out.append(f"acc[{g}] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[{widx}], bb[{p & 1}][{ks}], {src}, 0, 0, 0);  // slot {k}")
    if last_slot_of[(p, ks)] == k and p + 2 <= NSETS + CARRY - 1:
        out.append(frag(p + 2, ks))
    for kind, code in ops[k]:
        if kind == "store":
            out.append(f"if constexpr (!(ABL & 8)) {{ {code} }}")
        elif kind == "epi":
            out.append(code)
        elif kind == "load":
            out.append(f"if constexpr (!(ABL & 16)) {{ {code} }}")
        else:
            out.append(f"if constexpr (!(ABL & 1)) {{ {code} }}")
    out.append("NAF_SLOT_PIN;")
out.append("NAF_SCHED_TAIL_MARK;")
out += tail
out.append("#endif")

path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "naf_amd", "csrc", "stem_conv_sched3.inc")
if len(sys.argv) > 1:
    path = sys.argv[1]
with open(path, "w") as f:
    f.write("\n".join(out) + "\n")
busy = sum(1 for o in ops if o)
print(f"wrote {path}: order {ORDER}, {NSLOT} slots, {busy} carry side work, max micro-ops per slot {max(len(o) for o in ops)}")
