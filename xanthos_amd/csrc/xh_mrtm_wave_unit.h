// Device code shared by the two time-skewed dataflow routing kernels:
//   k_mrtm_wave (xh_mrtm_wave.hip)   every row sum in scipy's stored order: bit-identical to the reference, one
//                                    specialisation per row shape (pair units: every lane passes {F, F2})
//   k_mrtm_rsum (xh_mrtm_rsum.hip)   the reassociated ("tolerance") form: running sums along chains of lanes, two reads per
//                                    sub-step for every unit, the update fused (xh_flow_rsum.cpp plans it)
// wave_unit() is one unit's whole run; wave_claim() settles which unit a workgroup runs.  Included by exactly those two
// translation units (anonymous namespace: each has its own copy).
#pragma once
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <type_traits>

#include "xh_mrtm_flow.h"

namespace {


constexpr int LANES = 64;
constexpr int SK_P = 4;                       // row terms either side of the diagonal (D8: 4 smaller, 4 larger ids)
// End of round 4, both on: mrtm_route 23.3 - 23.6 -> 22.5 - 22.6 ms (same box, three alternating pairs; 0 switches either off
// for an A/B build: make exp EXPFLAGS="-DXH_WAVE_MIDZONE=0 -DXH_WAVE_BMOV=0").
#ifndef XH_WAVE_MIDZONE
#define XH_WAVE_MIDZONE 1   // a copy of the boundary loop for the zones that are neither the first nor the last of a run (see substep)
#endif
#ifndef XH_WAVE_BMOV
#define XH_WAVE_BMOV 1      // month-start snapshots as 64-bit moves under the crossing lanes' mask (see substep)
#endif
constexpr int NSLOT = 2 * LANES + 1;          // entries per LDS slot: cells, ghosts (imported streams), constant zero
constexpr int RING = 8;                       // LDS slots = sub-steps per stream block
constexpr int GROUP = 16;                     // sub-steps per unrolled group (two blocks)
constexpr int SK_R = 2;                       // block-transfer rounds: up to 8 * SK_R imports / outlets per unit
#ifndef XH_WAVE_CH
#define XH_WAVE_CH 256                        // (128 until round 4: 23.55 -> 23.35 ms at the full grid, same box, two runs each)
#endif
constexpr int CH = XH_WAVE_CH;                // iterations between flow-control checks (multiple of GROUP)
constexpr unsigned FAULT_DATA_WAIT = 1, FAULT_RING_WAIT = 2, FAULT_PLACE_WAIT = 3, FAULT_GUARD = XH_FAULT_GUARD;
constexpr unsigned FAULT_TEST = 99;
constexpr int PLACE_KEYS = 16 * 8 * 2 * 16 * 4;      // (xcc, se, sh, cu, simd) of HW_ID / XCC_ID

constexpr int PLACE_WORDS = 16 + PLACE_KEYS + PLACE_KEYS / 4;      // counters, one word per SIMD, one per CU

struct MonthRec {                             // one iteration of the schedule (spin-up months, then every month)
    int m, nt, g, write;                      // month index, sub-steps, first global sub-step, 1 = simulation pass
    double secs;
    long long q_off;                          // byte offset of month m inside a cell's row of the runoff source
};
static_assert(sizeof(MonthRec) == 32, "MonthRec is read with one s_load_dwordx8");

// What the month bookkeeping of iteration `it` needs, gathered in one record that is loaded a whole month before it is
// used (round 3 profile: three dependent loads of month records, each waited for, cost every unit ~30 cycles per sub-step)
struct FinRec {
    int m_prev_w, nt_prev;                    // month index (bit 30: write flag) and sub-steps of iteration it - 1
    int m_next2;                              // month index of iteration it + 2 (its runoff is loaded now)
    int g_next1;                              // first global sub-step of iteration it + 1
    double secs_next1;                        // seconds of iteration it + 1
    long long q_off_next2;                    // byte offset of month m_next2 inside a cell's row of the runoff source
};
constexpr int FIN_WRITE = 1 << 30;
static_assert(sizeof(FinRec) == 32, "FinRec is read with one s_load_dwordx8");

struct WaveArgs {
    const int *cell_of_slot, *lag, *ghost_lag, *export_edge, *ghost_edge, *edge_cons_unit;
    const unsigned *ent2;             // [2][SK_P][units*64] LDS entry offsets x 16 (before / after the diagonal)
    const unsigned *eprev;            // [units*64] chained units: entry x 16 of the pair this lane's own flows are added to
    const int *unit_p, *unit_lmax, *unit_glmax;
    const unsigned char *lane_flags;  // [units*64] single-sum plans (k_mrtm_rsum): bit 0 the cell may fire, bit 1 an exit lane (xh_flow_rsum.cpp)
    int64_t total_slots;
    int nmonths, nit, total;
    const int *unit_order;            // [units] claim list: units without streams by rising cost, then the others
    unsigned *place;                  // [PLACE_WORDS] counters of the placement, zeroed before every launch
    int n_units;
    int odd_ok;                       // 0: every month has an even number of sub-steps: lanes only cross a month start at even iterations
    const MonthRec *rec;              // [nit + 3], the last three zero
    const FinRec *fin;                // [nit + 2]
    double dt, dtinv;
    const double *flow_dist, *velocity, *area, *runoff, *S0;
    double *chs, *avg, *S_end, *F_end;
    // runoff source: a cell's row starts at runoff + cell * q_row_stride bytes, month m of it at the record's q_off.  The
    // [ncell, nmonths] array (stride nmonths * 8, q_off = m * 8), or the staged copy of a fed run (FlowFeed: stride 128,
    // q_off = (m / 16) * ncell * 128 + (m % 16) * 8) together with the months-ready word and the placement epoch.
    unsigned q_row_stride;
    unsigned ready_at_launch;         // months known final at launch (UINT_MAX: all of them, nothing to wait for)
    const unsigned *months_ready;
    unsigned *place_epoch;
    unsigned epoch;
    int n_excl;                       // k_mrtm_rsum, single-sum plans: the last n_excl units of unit_order (pair units) get a CU each to themselves
    int fenced;                       // XH_ROUTE_FENCED=1: agent-scope release / acquire fences around the stream counters (see check())
    char *xbuf;                       // [edges][RS] {F, F2}
    unsigned xbytes;                  // size of the rings
    unsigned ring_mask_b;             // RS * 16 - 1
    int rs;                           // RS
    unsigned *ready;                  // [edges] sub-steps published
    unsigned *done;                   // [units] sub-steps consumed
    unsigned *fault;
    unsigned long long *stats;
    const int *fold_cell;             // (at the end: k_mrtm_rsum only) [units*64] the leaf cell a lane carries besides its own, or -1; NULL: none
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Bound of one wait: 5 s of the 100 MHz real-time counter, two orders of magnitude above the kernel's run time at the
// full grid.  When it does expire (units of two dataflow kernels sharing the device) the call is re-routed with one
// workgroup per network (xh_fault_check).  A literal on purpose: passed in, it kept one more scalar live through the
// sub-step loop (round 2: 2-5 % in spills).
constexpr unsigned long long SPIN_LIMIT_TICKS = 500000000ull;

// Lanes with `need` wait until *p >= target (per lane); `seen` keeps the last value each lane read, so that the next
// check can skip the poll (a counter only grows).  False (and the fault word raised) on timeout / fault.
__device__ __forceinline__ bool wave_wait_ge(bool need, const unsigned *p, unsigned target, unsigned &seen,
                                             unsigned *fault, unsigned code) {
    bool ok = !need || seen >= target;
    if (__all(ok)) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (!ok) {
            seen = ld_relaxed(p);
            ok = seen >= target;
        }
        if (__all(ok)) return true;
        if (ld_relaxed(fault) != 0) return false;
        if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT_TICKS) {
            __hip_atomic_store(fault, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(4);
    }
}

typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;                   // raw-buffer cache policy: agent-coherent (write-through / re-fetching)
typedef __attribute__((address_space(3))) const char lds_cchar;
typedef __attribute__((address_space(3))) char lds_char;

// The arguments live in device memory and every use re-reads the field it needs (scalar loads of a laundered pointer):
// held as kernel arguments, the ~70 scalar registers of pointers and sizes stayed live across the sub-step loop and were
// spilled to vector lanes and restored around every group of 16 sub-steps (round 2).
typedef __attribute__((address_space(4))) const WaveArgs WaveArgsK;      // constant address space: always scalar loads
template <class T>
__device__ __forceinline__ T xh_ldarg(__attribute__((address_space(4))) const T *p) {
    asm volatile("" : "+s"(p));
    return *p;
}
#define A(f) xh_ldarg(&ap->f)

// value exchanged between lanes: the pair {F, F2} (or {sum F, sum F2}), or the one running sum of a single unit (SGL = 1)
template <bool ONE> struct Val;
template <> struct Val<false> {
    typedef v2d T;
    typedef __attribute__((address_space(3))) const v2d lds_c;
    typedef __attribute__((address_space(3))) v2d lds_m;
    static constexpr unsigned B = 16;
    static __device__ __forceinline__ T zero() { return v2d{0.0, 0.0}; }
};
template <> struct Val<true> {
    typedef double T;
    typedef __attribute__((address_space(3))) const double lds_c;
    typedef __attribute__((address_space(3))) double lds_m;
    static constexpr unsigned B = 8;
    static __device__ __forceinline__ T zero() { return 0.0; }
};

// NG: rounds of 8 imported streams the unit takes part in (0, 1, 2).  Most units with imports have at most 8: a second
// round that loads nothing still costs its load, its LDS store and its address arithmetic in every block of 8 sub-steps.
// RSUM: the reassociated form (k_mrtm_rsum; tables of xh_flow_rsum.cpp).  PRE = 1 / 0: the unit reads its cells' inflow -- the
// running sum {sum F, sum F2} over ALL upstream neighbours of a cell, left by the last member of their chain -- or holds
// only cells without any; CHAIN: it reads a cell's predecessor in its own chain; POST = 0.  The update of mrtm.py:50-69 is
// fused: with a = 1 - tauinv dt and the lateral inflow pre-multiplied by dt,
//     base = S a + erl dt           S1 = base + (sum F) dt     -- the trial storage; "sx" (mrtm.py:54) is S1 < 0
//     F2   = sx ? F + S1 / dt : F   (mrtm.py:60: inbound + lateral + S / dt = F + S1 / dt)
//     S    = sx ? 0 : base + (sum F2) dt                        (mrtm.py:63, 69)
// -- 8 fp64 operations, one compare, four selects instead of ~20 + 2 per row term; equal to the reference's sequence to
// rounding (tested to 1e-9 of every routed value against the oracle at the full grid; the gate is 1e-6), not bit for bit.
// FOLD (with RSUM, units without streams only): a lane also carries ONE leaf child of its cell that does not fire
// (velocity * dt / length < 1 and a storage that never turns negative: guarded) -- for such a cell the step of mrtm.py:50-69
// is the linear recurrence S <- S a + erl dt with outflow S / tau, which costs the lane four fp64 operations per sub-step
// (outflow, recurrence, the month's flow sum, one fma into the carrier's `base`) instead of a lane, an LDS slot and a level
// of lag of its own.  The leaf runs at the carrier's own sub-step; its month bookkeeping rides on the carrier's.
//
// The guard.  Within a month erl is constant and x -> fma(x, a, erl dt) is non-decreasing in x, so the storages of a month
// form a monotone sequence (in floating point too): if the month starts and ends at S >= 0 no sub-step in between had S < 0,
// and the reference's branch mrtm.py:54 was never taken -- the recurrence IS the reference's step.  The kernel therefore
// looks at the storage once per month (FOLD_EPS below); a month that ends lower makes the unit give up (FAULT_GUARD: the
// host routes the call again on the plan without folded leaves and stays on it).  Runoff that is negative by a rounding
// error of the runoff model (-1e-17 mm on a dry cell) takes the storage of an empty leaf to about -1e-11 m3: the reference
// passes that volume on at once (mrtm.py:60, a flow of -1e-15 m3/s), the recurrence over 1 / (tau^-1 dt) sub-steps; both
// conserve it, and the difference is nine orders below this form's bar (1e-9 m3/s, 1e-3 m3: xh_mrtm.hip k_count_far).
// FOLD_EPS = 1e-6 m3 keeps the guard from tripping on that while bounding what it lets through to 1e-10 m3/s.
//
// SGL (with RSUM; single-sum plans of xh_flow_rsum.cpp, whose header has the argument).  1: a SINGLE unit -- none of its cells
// may fire AND have an upstream neighbour that may, so the two sums of mrtm.py:51 and :66 are the same number whenever they
// matter, and the lanes pass ONE running sum, that of the adjusted flows F2, in 8-byte entries (the layout of the plain units:
// half the LDS traffic); the step is
//     S1 = base + (sum F2) dt,   m = min(S1, 0),   F2 = F + m / dt,   S = S1 - m
// (S = 0 exactly where the cell fires, S1 where it does not, NaN where S1 is NaN: min returns the 0).  Outlets export {y, y},
// imports take the second half of a pair.  2: a PAIR unit of such a plan -- the few cells that do need both sums: the step
// of SGL = 0, plus the guard of its exit lanes.  Guards (a trip makes the host route the call again on the plan of pairs):
// the plan was made for the cells that can fire at THIS velocity, length and dt (a cell that can but is not marked trips
// it), initial storages and -- month by month -- lateral inflows are >= 0 up to a rounding error of the runoff model, and
// the outflow of every exit lane (lane_flags bit 1: a cell in pair form whose downstream cell is not) stays >= -SGL_XEPS:
// then every flow a single unit receives is >= 0, and its cells outside the marked set cannot fire.
template <int PRE, int POST, int NG, bool CHAIN, bool RSUM = false, bool FOLD = false, int SGL = 0>
__device__ __forceinline__ void wave_unit(WaveArgsK *ap, char *lds_generic, uint2 *xtab,
                                          __attribute__((address_space(3))) unsigned *qstage,
                                          __attribute__((address_space(3))) double *fend, const int unit) {
    constexpr bool HAS_G = NG > 0;
    constexpr bool V8 = SGL == 1;                      // 8-byte entries: one value per lane instead of a pair
    typedef Val<V8> V;
    typedef typename V::T val_t;
    typedef typename V::lds_c lds_cv;
    typedef typename V::lds_m lds_mv;
    constexpr unsigned VB = V::B;
    constexpr unsigned SLOTB = NSLOT * VB;                 // bytes per ring slot
    lds_char *lds0 = (lds_char *)lds_generic;
    const int lane = threadIdx.x;
    const int64_t slot = (int64_t)unit * LANES + lane;

    const int gc = A(cell_of_slot)[slot];
    const bool valid = gc >= 0;
    const int gc_safe = valid ? gc : 0;      // idle lanes load cell 0's runoff, and ignore it
    // byte offset of this lane's row in the [ncell, nmonths] arrays: a 32-bit register that is never redefined, so that the
    // monthly loads / stores address memory as (uniform base + this) and no address register of an access in flight is ever
    // overwritten (the compiler answers that with s_waitcnt vmcnt(0): ~2 us per month behind the output stores)
    const unsigned row_off = (unsigned)gc_safe * A(q_row_stride);
    const double tauinv = valid ? A(velocity)[gc] / A(flow_dist)[gc] : 0.0;      // mrtm.py:40
    const double acoef = 1.0 - tauinv * A(dt);                                    // RSUM: share of the storage a sub-step keeps
    const double area = valid ? A(area)[gc] : 0.0;
    const double S0v = (valid && A(S0)) ? A(S0)[gc] : 0.0;
    // table offsets are entry x 16 (the pair layout); a single unit's entries are 8 bytes
    auto ent_off = [](unsigned o) { return V8 ? o >> 1 : o; };
    static_assert(!RSUM || (PRE <= 1 && POST == 0), "reassociated form: one inflow entry");
    static_assert(!FOLD || (RSUM && NG == 0), "folded leaves: reassociated form, units without imports");
    static_assert(SGL == 0 || RSUM, "single-sum plans: reassociated form");
    // the folded leaf of this lane
    const int gcL = FOLD ? A(fold_cell)[slot] : -1;
    const bool validL = FOLD && valid && gcL >= 0;
    const unsigned row_offL = (unsigned)(validL ? gcL : 0) * A(q_row_stride);
    const double tauL = validL ? A(velocity)[gcL] / A(flow_dist)[gcL] : 0.0;
    const double acoefL = 1.0 - tauL * A(dt);
    const double areaL = validL ? A(area)[gcL] : 0.0;
    const double S0L = (validL && A(S0)) ? A(S0)[gcL] : 0.0;
    double SL = 0.0, favgL = 0.0, erlL = 0.0, erlL_n = 0.0, snapSL = 0.0, snapAL = 0.0, FL = 0.0;
    // guard of the folded leaves (above): velocity * dt / length not below 1 (the plan was made for other data), a negative
    // or NaN ratio, a negative initial storage -- and, month by month in finalize(), a storage below -FOLD_EPS
    constexpr double FOLD_EPS = 1e-6;
    unsigned gfold = (validL && !(tauL * A(dt) <= 1.0 - 1.0 / 1048576.0 && tauL >= 0.0 && S0L >= 0.0)) ? 1u : 0u;
    // guards of a single-sum plan (SGL; the argument is in front of the template)
    constexpr double SGL_EPS = 1e-6;       // m3 per sub-step: lateral inflow below this trips the guard (the runoff model leaves -1e-17 mm on dry cells)
    constexpr double SGL_XEPS = 1e-10;     // m3/s: outflow of an exit lane
    unsigned gsgl = 0;
    bool xlane = false;
    double fmin_seen = 0.0;                // SGL = 2: smallest outflow of this lane so far
    if (SGL != 0) {
        const unsigned lf = A(lane_flags)[slot];
        xlane = valid && (lf & 2u) != 0;
        gsgl = (valid && (!(tauinv >= 0.0) || (!(tauinv * A(dt) <= 1.0 - 1.0 / 1048576.0) && !(lf & 1u)) || S0v < 0.0)) ? 1u : 0u;
    }
    constexpr int PRE_N = PRE > 0 ? PRE : 1, POST_N = POST > 0 ? POST : 1;      // (array extents: a side may be empty)
    lds_cchar *epre[PRE_N], *epost[POST_N];
#pragma unroll
    for (int w = 0; w < PRE; ++w) epre[w] = lds0 + ent_off(A(ent2)[(int64_t)w * A(total_slots) + slot]);
#pragma unroll
    for (int w = 0; w < POST; ++w) epost[w] = lds0 + ent_off(A(ent2)[(int64_t)(SK_P + w) * A(total_slots) + slot]);
    // CHAIN: the cells that feed a cell from in front of its diagonal form a chain in their stored order; each runs one
    // level behind the one before it and stores {running sum + F, running sum + F2} instead of {F, F2}; the cell they
    // feed reads the last one's value as ONE term.  The additions and their order are the row sum's own.
    lds_cchar *eprv = lds0 + (CHAIN ? ent_off(A(eprev)[slot]) : 0u);
    lds_mv *own = (lds_mv *)lds0 + lane;
    const int xedge = A(export_edge)[slot];
    const int gedge = A(ghost_edge)[slot];
    const bool has_x = xedge >= 0, has_g = gedge >= 0;
    const unsigned long long xmask = __ballot(has_x), gmask = __ballot(has_g);
    const bool any_x = xmask != 0, any_g = HAS_G;
    const int nx_out = __popcll(xmask), ng = __popcll(gmask);       // outlets / imports of this unit
    const int lmax = A(unit_lmax)[unit], glmax = A(unit_glmax)[unit];
    const int lag_g = has_g ? A(ghost_lag)[slot] : 0;
    const unsigned *ready_p = A(ready) + (has_g ? gedge : 0);
    const unsigned *done_p = A(done) + (has_x ? A(edge_cons_unit)[xedge] : 0);
    const unsigned maskb = A(ring_mask_b);
    const unsigned oob = ~maskb;      // ring offset of an idle block lane: beyond the rings (wave_launch checks), dropped by the hardware
    // Streams are read and written through a buffer resource with the agent-coherent cache policy: stores write through
    // to memory (whole 128-byte lines per outlet and block), loads re-fetch.  No release / acquire fence around the
    // counters (an agent-scope release writes back the whole L2 of the XCD: -10 %); see check().
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(A(xbuf), 0, (int)A(xbytes), 0x00020000);
    const int total = A(total), nit = A(nit);
    const bool odd_ok = A(odd_ok) != 0;

    // ---- block-transfer roles: lane (k = lane / 8 + 8 r, i = lane % 8) moves sub-step i of outlet / import k.
    //      No exec masks: an idle block lane (k beyond the unit's imports / outlets) loads from and stores to ring offset
    //      `oob`, which the buffer resource answers with zeros / drops, and writes ghost entry k, which no row refers to.
    if (has_x) xtab[__popcll(xmask & ((1ull << lane) - 1ull))] = make_uint2((unsigned)lane, (unsigned)xedge);
    const int sub = lane & 7, grp = lane >> 3;
    const bool x2 = nx_out > 8;                          // a second round of outlet stores is needed (uniform)
    unsigned gfull[SK_R], xbyte[SK_R];      // ring base | position of the next import block; ring base + 16 i for stores
    lds_mv *gdst[SK_R];
    lds_cv *xsrc[SK_R];
#pragma unroll
    for (int r = 0; r < SK_R; ++r) {
        const int k = r * 8 + grp;
        const bool gon = k < ng, xon = k < nx_out;
        const int ge = gon ? A(ghost_edge)[(int64_t)unit * LANES + k] : 0;
        const int gl = gon ? A(ghost_lag)[(int64_t)unit * LANES + k] : 0;
        // position of sub-step (sub - lag) of block 0; advanced by 128 bytes per block inside the ring
        gfull[r] = gon ? ((unsigned)ge * (maskb + 1u)) | (((unsigned)(sub - gl) * 16u) & maskb) : oob;
        gdst[r] = (lds_mv *)(lds0 + sub * SLOTB) + LANES + (k & 63);
        const uint2 t = xon ? xtab[k] : make_uint2((unsigned)lane, 0u);
        xbyte[r] = xon ? t.y * (maskb + 1u) + (unsigned)sub * 16u : oob;
        xsrc[r] = (lds_cv *)(lds0 + sub * SLOTB) + t.x;
    }
#pragma unroll
    for (int k = 0; k < RING; ++k) {
        own[k * NSLOT] = V::zero();
        own[k * NSLOT + LANES] = V::zero();
        if (lane == 0) own[k * NSLOT + 2 * LANES] = V::zero();
    }

    const double dt = A(dt), dtinv = A(dtinv);
    double S = 0.0, F = 0.0, favg = 0.0, erl = 0.0;
    double snapS = 0.0, snapA = 0.0;
    int nx = A(lag)[slot];                                  // iteration at which this lane enters its next month
    // pointers the month bookkeeping needs, read once (nine scalar registers; the sub-step loop holds no scalar loads)
    const double *p_runoff = A(runoff);
    double *p_chs = A(chs), *p_avg = A(avg);
    typedef __attribute__((address_space(4))) const MonthRec MonthRecK;      // scalar loads, whatever the kernel stores elsewhere
    MonthRecK *p_rec = (MonthRecK *)A(rec);
    typedef __attribute__((address_space(4))) const FinRec FinRecK;
    FinRecK *p_fin = (FinRecK *)A(fin);
    auto ld_fin = [&](int i) {
        FinRec r;
        r.m_prev_w = p_fin[i].m_prev_w;
        r.nt_prev = p_fin[i].nt_prev;
        r.m_next2 = p_fin[i].m_next2;
        r.g_next1 = p_fin[i].g_next1;
        r.secs_next1 = p_fin[i].secs_next1;
        r.q_off_next2 = p_fin[i].q_off_next2;
        return r;
    };
    FinRec fc = ld_fin(0);               // record of the next month bookkeeping, loaded one month ahead
    auto ld_rec = [&](int i) {
        MonthRec r;
        r.m = p_rec[i].m;
        r.nt = p_rec[i].nt;
        r.g = p_rec[i].g;
        r.write = p_rec[i].write;
        r.secs = p_rec[i].secs;
        r.q_off = p_rec[i].q_off;
        return r;
    };
    const int nmo = A(nmonths);
    // The runoff of the month after next travels global memory -> LDS without a register (global_load_lds_dword, lane L's
    // dword lands at M0 + instruction offset + 4 L: tools/micro/lds_dma.hip) and is read out of LDS a month later.  As an
    // ordinary load its result was a register in flight across the sub-step loop, and the compiler answered that with
    // s_waitcnt vmcnt(0) in front of the next sub-step -- behind the month's output stores, ~2 us per month and unit
    // (round 3 profile: ~30 cycles per sub-step of every unit, whatever the order of loads and stores).
    const unsigned q_lds = (unsigned)(size_t)qstage;
    auto runoff_fetch = [&](long long q_off) {       // asynchronous; complete before the next month bookkeeping (see runoff_take)
        const char *src = reinterpret_cast<const char *>(p_runoff) + q_off + (size_t)row_off;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dword %1, off\n\ts_add_u32 m0, m0, 252\n\t"
                     "global_load_lds_dword %1, off offset:4\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(q_lds)
                     : "memory");
        if (FOLD) {      // the folded leaves' rows: a second staging area of 2 x 64 dwords behind the first
            const char *srcL = reinterpret_cast<const char *>(p_runoff) + q_off + (size_t)row_offL;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dword %1, off\n\ts_add_u32 m0, m0, 252\n\t"
                         "global_load_lds_dword %1, off offset:4\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(srcL), "s"(q_lds + 2u * LANES * 4u)
                         : "memory");
        }
    };
    auto runoff_take = [&]() {
        // vmcnt retires in order.  A unit with streams has issued at least one stream access per block of 8 sub-steps since
        // the fetch (>= 6 blocks: a month is at least lmax + 32 >= 48 sub-steps), so "at most 4 still in flight" covers the
        // fetch without draining the import loads; a unit without streams has nothing else in flight.
        if (any_x || any_g) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned lo = qstage[lane], hi = qstage[LANES + lane];
        return __hiloint2double((int)hi, (int)lo);
    };
    double erl_n = 0.0;                                    // lateral inflow of the month to enter
    {
        const MonthRec r0 = ld_rec(0), r1 = ld_rec(1);
        auto ld_q = [&](long long q_off) {
            return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(p_runoff) + q_off + (size_t)row_off);
        };
        const double q0 = ld_q(r0.q_off);
        erl_n = ((valid ? q0 : 0.0) * area) * 1000.0 / r0.secs;                  // mrtm.py:45
        if (RSUM) erl_n *= A(dt);
        if (SGL != 0) gsgl |= (erl_n < -SGL_EPS) ? 2u : 0u;
        if (FOLD) {
            const double q0L = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(p_runoff) + r0.q_off + (size_t)row_offL);
            erlL_n = (((validL ? q0L : 0.0) * areaL) * 1000.0 / r0.secs) * A(dt);
        }
        if (nit > 1) runoff_fetch(r1.q_off);
    }
    // month outputs leave as groups of OB months per cell (32 bytes = one memory sector)
    constexpr int OB = 4;               // (groups of 2 free eight registers and cost 1 - 2 %: round 4, same-box A/B)
    double ob_s[OB], ob_a[OB];
    double ob_sL[FOLD ? OB : 1], ob_aL[FOLD ? OB : 1];      // the folded leaf's months
#pragma unroll
    for (int j = 0; j < OB; ++j) ob_s[j] = ob_a[j] = 0.0;
#pragma unroll
    for (int j = 0; j < (FOLD ? OB : 1); ++j) ob_sL[j] = ob_aL[j] = 0.0;
    bool alive = true;
    // Fed run (FlowFeed): months [0, mready) of the runoff source are known to be final.  A month beyond that is waited
    // for, bounded like every wait here, on the months-ready word (written by a kernel that runs after the one that
    // produced the months; read like the stream counters).  The staged source is laid out so that no line is ever read
    // before all of it is final, hence no invalidate between the flag and the data.
    unsigned mready = A(ready_at_launch);
    auto wait_months = [&](unsigned need) {
        const unsigned *p = A(months_ready);
        if (!p) return true;                             // (cannot happen: ready_at_launch covers the series then)
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            // An atomic read-modify-write (+ 0), not a load: it is carried out where the word lives, so it cannot be answered
            // from a copy of the line that an earlier poll left in this XCD's L2 (a unit that polls while every other unit is
            // parked in the same wait has no traffic that would ever evict such a copy).  Rare path: once per call and unit.
            unsigned v = 0;
            if (lane == 0) v = __hip_atomic_fetch_add(const_cast<unsigned *>(p), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
            if (v >= need) {
                mready = v;
                return true;
            }
            if (ld_relaxed(A(fault)) != 0) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT_TICKS) {
                __hip_atomic_store(A(fault), FAULT_DATA_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
            __builtin_amdgcn_s_sleep(16);
        }
    };
    unsigned long long cyc_wait_data = 0, cyc_wait_ring = 0, zone_groups = 0;
    const unsigned long long cyc_begin = __builtin_amdgcn_s_memtime();
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();

    // ---- flow control, every CH iterations, at a group start.  Every block of 8 iterations issues at least one
    //      stream access (the round-4 form of the publication counted on that: "all but the 8 youngest memory operations
    //      have completed" + a lag of 64 iterations).
    //      MEMORY-ORDERING ASSUMPTION (outside the HIP memory model, stated here because everything rests on it): the
    //      stream stores are write-through `sc1` buffer stores, and a write-through store retires (leaves vmcnt) only
    //      when the memory side has acknowledged it; since round 5 the publication waits for `s_waitcnt vmcnt(0)` -- ALL
    //      of the wave's memory operations, the form the guide measured for `sc1` hand-offs -- so every stream store
    //      issued so far is visible at agent scope; only then is the
    //      counter advanced (relaxed agent-scope store, itself ordered behind the waitcnt by the "memory" clobber).  (Until
    //      round 4: vmcnt(8) + the in-order retirement of vmcnt + a publication lag of 64 iterations.)  A
    //      consumer reads the counter with an agent-scope load and the data with `sc1` loads, which re-fetch past its
    //      XCD's L2.  No release / acquire fences.  Evidence: every full-size launch of the test suite bit-identical to
    //      the oracle, XH_ROUTE_VALIDATE (the same call routed by the barrier-only kernel and compared on the device),
    //      two contexts routing concurrently, the fuzzers; a violation would show as a wrong bit, a lost wake-up as a
    //      bounded-wait fault.  Each check asks for the counters the NEXT check will look at (pend_*, loaded by inline
    //      asm so that no wait is attached to them) and first looks at the values asked for 128 iterations ago; only if
    //      those do not cover its needs does it poll.  Counters only grow, so a stale value is merely conservative.
    unsigned seen_ready = 0, seen_done = 0, pend_ready = 0, pend_done = 0;
    auto load_async = [&](const unsigned *q) {
        unsigned v;
        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(q) : "memory");
        return v;
    };
    auto check = [&](int n) {
        const unsigned long long w0 = __builtin_amdgcn_s_memtime();
        // Round 5, the default since: EVERY memory operation of the wave has been acknowledged before the counter
        // store -- `s_waitcnt vmcnt(0)` in front of the flag, the guide's measured `sc1` hand-off form, without the L2 write-back
        // of a release fence; the import loads two blocks ahead are simply waited for.  Priced on both kernels, same box,
        // alternating (profiles/round5/fence_mid_ab.txt): 15.08 - 15.38 against 15.16 - 15.36 ms (reassociated), 22.0 - 22.9
        // against 22.3 - 23.0 (bit-exact) -- nothing; the full release / acquire pair costs +76 % / +40 % (XH_ROUTE_FENCED=1).
        // (Until round 4: all but the 8 youngest operations + a publication lag of 64 sub-steps.)
        // (The second wait is a statement of its own BEHIND the counted one, without register operands: pend_* are registers
        // of loads in flight until that wait, and anything that makes the compiler copy them first -- a branch around the wait
        // did, and every stream-linked unit then read counters that had not arrived -- reads garbage.)
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(pend_ready), "+v"(pend_done) : : "memory");   // older stores acknowledged, pend_* in
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (SGL == 2) gsgl |= (xlane && fmin_seen < -SGL_XEPS) ? 4u : 0u;
        if ((FOLD && __any(gfold != 0)) || (SGL != 0 && __any(gsgl != 0))) {      // a folded leaf that can fire after all, a single-sum plan made for other data: the host routes again on the plan of pairs
            __hip_atomic_store(A(fault), FAULT_GUARD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            alive = false;
        }
        const bool fenced = A(fenced) == 1;
        if (any_x) {      // publish what has certainly been stored, then make sure the next CH iterations have ring space
            // (everything stored so far is acknowledged: publish it all)
            const int pub = min(max(n - RING - lmax, 0), total);
            // XH_ROUTE_FENCED=1: the publication the HIP memory model asks for -- an agent-scope release (buffer_wbl2 sc1 +
            // s_waitcnt vmcnt(0): every memory operation of the wave drained, the XCD's L2 written back) in front of the
            // counter store, an agent-scope acquire behind the consumer's counter load.  Measured on MI355X at the full grid
            // (profiles/round4/fenced_ab.txt); the default keeps the vmcnt(8) form above and the first call of a plan on a
            // new box / build is cross-checked against the barrier-only kernel instead (xh_mrtm.hip, first_check_*).
            if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (has_x) __hip_atomic_store(A(ready) + xedge, (unsigned)pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int need = n + CH - lmax - A(rs);
            seen_done = max(seen_done, pend_done);
            if (need > 0 && alive)
                alive = wave_wait_ge(has_x, done_p, (unsigned)min(need, total), seen_done, A(fault), FAULT_RING_WAIT);
        }
        const unsigned long long w1 = __builtin_amdgcn_s_memtime();
        if (any_g && alive) {   // the next CH iterations load up to sub-step n + CH + GROUP - 1 - lag_g
            if (lane == 0)    // every import has been consumed up to n - glmax
                __hip_atomic_store(A(done) + unit, (unsigned)min(max(n - glmax, 0), total), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            const int need = min(total, n + CH + GROUP - lag_g);
            seen_ready = max(seen_ready, pend_ready);
            alive = wave_wait_ge(has_g && need > 0, ready_p, (unsigned)max(need, 0), seen_ready, A(fault), FAULT_DATA_WAIT);
            if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("" ::: "memory");      // the stream loads stay behind the poll
        }
        if (any_x) pend_done = load_async(done_p);
        if (any_g) pend_ready = load_async(ready_p);
        cyc_wait_ring += w1 - w0;
        cyc_wait_data += __builtin_amdgcn_s_memtime() - w1;
    };

    // ---- month bookkeeping for all lanes at once: outputs of iteration it - 1, lateral inflow of iteration it + 1.
    //      One 32-byte record per month involved (rec has three zero records past the end: no bounds tests).
    auto finalize = [&](int it) {
        const FinRec f = fc;
        fc = ld_fin(it + 1 <= nit ? it + 1 : nit + 1);      // used a month from now: nobody waits for it
        // The runoff loaded a month ago is consumed BEFORE this month's output stores are issued: after them, the wait for
        // it would also wait for the stores (vmcnt counts in order), ~2 us per month and unit (round 3 profile: ~30 cycles
        // per sub-step of every unit).
        if (it + 1 < nit) {
            const double qn = runoff_take();
            erl_n = ((valid ? qn : 0.0) * area) * 1000.0 / f.secs_next1;
            if (RSUM) erl_n *= dt;
            if (SGL != 0) gsgl |= (erl_n < -SGL_EPS) ? 2u : 0u;
            if (FOLD) {      // (runoff_take has waited for the loads of both staging areas: a unit without streams waits vmcnt(0))
                const unsigned lo = qstage[2 * LANES + lane], hi = qstage[3 * LANES + lane];
                const double qnL = __hiloint2double((int)hi, (int)lo);
                erlL_n = (((validL ? qnL : 0.0) * areaL) * 1000.0 / f.secs_next1) * dt;
            }
        }
        if (it >= 1) {
            const int m = f.m_prev_w & (FIN_WRITE - 1);
            const bool write_prev = (f.m_prev_w & FIN_WRITE) != 0;
#pragma unroll
            for (int j = 0; j < OB - 1; ++j) {
                ob_s[j] = ob_s[j + 1];
                ob_a[j] = ob_a[j + 1];
            }
            ob_s[OB - 1] = snapS;
            ob_a[OB - 1] = snapA / (double)f.nt_prev;                          // mrtm.py:80
            if (FOLD) {
#pragma unroll
                for (int j = 0; j < OB - 1; ++j) {
                    ob_sL[j] = ob_sL[j + 1];
                    ob_aL[j] = ob_aL[j + 1];
                }
                ob_sL[OB - 1] = snapSL;
                ob_aL[OB - 1] = snapAL / (double)f.nt_prev;
                gfold |= (snapSL < -FOLD_EPS) ? 2u : 0u;                       // (NaN storage: as the reference, not a firing)
                if (write_prev && validL) {
                    if ((m & (OB - 1)) == OB - 1) {
                        const int64_t o = (int64_t)gcL * nmo + (m - (OB - 1));
#pragma unroll
                        for (int j = 0; j < OB; j += 2) {
                            if (p_chs) *reinterpret_cast<v2d *>(p_chs + o + j) = v2d{ob_sL[j], ob_sL[j + 1]};
                            if (p_avg) *reinterpret_cast<v2d *>(p_avg + o + j) = v2d{ob_aL[j], ob_aL[j + 1]};
                        }
                    } else if (m == nmo - 1) {
                        const int r = (m & (OB - 1)) + 1;
                        const int64_t o = (int64_t)gcL * nmo + (m + 1 - r);
#pragma unroll
                        for (int j = 0; j < OB; ++j)
                            if (j >= OB - r) {
                                if (p_chs) p_chs[o + j - (OB - r)] = ob_sL[j];
                                if (p_avg) p_avg[o + j - (OB - r)] = ob_aL[j];
                            }
                    }
                }
            }
            if (write_prev && valid) {     // whole groups of OB months per cell
                if ((m & (OB - 1)) == OB - 1) {
                    const int64_t o = (int64_t)gc * nmo + (m - (OB - 1));
#pragma unroll
                    for (int j = 0; j < OB; j += 2) {
                        if (p_chs) *reinterpret_cast<v2d *>(p_chs + o + j) = v2d{ob_s[j], ob_s[j + 1]};
                        if (p_avg) *reinterpret_cast<v2d *>(p_avg + o + j) = v2d{ob_a[j], ob_a[j + 1]};
                    }
                } else if (m == nmo - 1) {                                      // last, partial group
                    const int r = (m & (OB - 1)) + 1;
                    const int64_t o = (int64_t)gc * nmo + (m + 1 - r);
#pragma unroll
                    for (int j = 0; j < OB; ++j)
                        if (j >= OB - r) {
                            if (p_chs) p_chs[o + j - (OB - r)] = ob_s[j];
                            if (p_avg) p_avg[o + j - (OB - r)] = ob_a[j];
                        }
                }
            }
        }
        if (it + 2 < nit) {                             // after runoff_take: one staging area
            // Fed run (FlowFeed): the month may not exist yet.  `mready` is the last value of the months-ready word this wave
            // saw (all ones when the series was complete at launch: the comparison is all an ordinary run pays, once a month).
            if ((unsigned)f.m_next2 >= mready) alive = wait_months((unsigned)f.m_next2 + 1u);
            if (alive) runoff_fetch(f.q_off_next2);
        }
        return f.g_next1;
    };

    // gathered values of the current sub-step (issued one iteration ago); import blocks in flight (two ahead)
    // Two register sets that swap roles every sub-step (set j & 1 is consumed, the other one is being read into): with
    // "current" and "next" variables copied at the end of each sub-step, the values in flight at the loop's back-edge did
    // not sit in the registers the loop header expects, and the compiler moved them there -- behind s_waitcnt vmcnt(0) and
    // lgkmcnt waits, i.e. every group of 16 sub-steps drained the import loads it had just issued (round 3 profile).
    val_t va[2][PRE_N], vb[2][POST_N], vr[2];
    v4u gbuf[2][SK_R];                   // imported pairs as raw words (see import_drop)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int w = 0; w < PRE; ++w) va[q][w] = V::zero();
#pragma unroll
        for (int w = 0; w < POST; ++w) vb[q][w] = V::zero();
        vr[q] = V::zero();                // CHAIN: running value of the cell before this one
    }
    auto import_load = [&](int r) {      // the block at gfull[r]; the position then moves on by one block
        const v4u v = __builtin_amdgcn_raw_buffer_load_b128(xr, gfull[r], 0, AUX_SC1);
        gfull[r] = (gfull[r] & ~maskb) | ((gfull[r] + RING * 16u) & maskb);
        return v;
    };
    auto import_drop = [&](int r, const v4u u) {
        if (SGL == 1) {      // a single unit takes the sum of the adjusted flows (a single producer exports {y, y})
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            *reinterpret_cast<__attribute__((address_space(3))) v2u *>(gdst[r]) = v2u{u.z, u.w};
        } else {
            *reinterpret_cast<__attribute__((address_space(3))) v4u *>(gdst[r]) = u;
        }
    };
    // Block work of the sub-steps m0 = 0 (mod 8), after their gather reads: drop the import block of iterations
    // m0 .. m0 + 7 into the ghost entries, load the block two ahead, read the outlets' block of m0 - 8 .. m0 - 1 out of
    // the LDS ring (before the end of this sub-step overwrites slot 0).  The block is stored one sub-step later
    // (block_store), behind that sub-step's counted wait: stored at once, the wave would sit out the LDS latency.
    val_t xb[SK_R];
#pragma unroll
    for (int r = 0; r < SK_R; ++r) {
        xb[r] = V::zero();
    }
    auto block_io = [&](const int b) {
        if (HAS_G) {      // the unit's rounds, unconditionally (NG is a template argument): with a run-time branch around the
                          // second one the compiler can no longer count the loads in flight and waits for ALL of them
                          // (vmcnt(0)) at every block: +60-100 cycles per sub-step
#pragma unroll
            for (int r = 0; r < NG; ++r) {
                import_drop(r, gbuf[b][r]);
                gbuf[b][r] = import_load(r);
            }
        }
        if (any_x) {
            xb[0] = *xsrc[0];
            if (x2) xb[1] = *xsrc[1];
        }
    };
    auto store_pair = [&](const val_t v, unsigned voff, unsigned soff) {
        v2d p;
        if constexpr (SGL == 1) p = v2d{v, v};      // a single unit's running sum: read as F = F2 by a pair unit (exact: see the template's comment)
        else p = v;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, p), xr, voff, soff, AUX_SC1);
    };
    auto block_store = [&](int m0) {
        if (any_x) {
            const unsigned xpos = ((unsigned)(m0 - RING - lmax) * 16u) & maskb;   // 8 sub-steps, never wrapping
            store_pair(xb[0], xbyte[0], xpos);
            if (x2) store_pair(xb[1], xbyte[1], xpos);
        }
    };

    check(0);
#pragma unroll
    for (int r = 0; r < SK_R; ++r) gbuf[0][r] = r < NG ? import_load(r) : v4u{0u, 0u, 0u, 0u};      // blocks of iterations 0..7
#pragma unroll
    for (int r = 0; r < SK_R; ++r) gbuf[1][r] = r < NG ? import_load(r) : v4u{0u, 0u, 0u, 0u};      // and 8..15

    const int N = (total + lmax + 1 + GROUP - 1) & ~(GROUP - 1);
    int itz = 0, gz = 0;                 // month whose start zone [gz, gz + lmax] is next (itz == nit: the end zone)
    int itf = 0, nf = (lmax + 1 + GROUP - 1) & ~(GROUP - 1);     // next month bookkeeping and its iteration
    int ntz = nit > 0 ? p_rec[0].nt : 0;

    // zone: some lanes cross a month start in this group; first: it is the start of the series (the lanes pick up S0)
    // edge: 1 = the start of the series (the lanes pick up S0), 2 = its end (the lanes leave their last outflow for F_end in
    // LDS: as a register selected at every month start like snapS it cost 0.3 ms of the full grid's 23.4 -- two more live
    // registers through every loop and two selects per boundary sub-step, for a value only the last month needs)
    // mid_c: the zone is neither the first nor the last of the series and lags and months are even (dt = 3 h) -- known at
    // compile time in a copy of the boundary loop of its own, so that all but two zones of a run carry neither the first
    // zone's pick-up of S0, nor the last zone's store of the outflow (an LDS store under an empty mask in every even
    // sub-step), nor a branch on odd_ok in every odd one.
    auto substep = [&](auto zone_c, auto mid_c, const int n, const int j, const int edge, const int rel) {
        // (a compile-time flag: as a run-time argument the optimiser folded the two variants of the group back into one
        // body with a branch around the boundary code in every sub-step)
        constexpr bool MID = decltype(mid_c)::value;
        if (decltype(zone_c)::value && ((j & 1) == 0 || (!MID && odd_ok))) {
            // The lanes whose lag puts them on the month start at this iteration: branch-free (a branch per lane set cost
            // 15 instructions per sub-step of the boundary groups and made the compiler copy the gathered values around)
            const bool c = rel == j;
#if XH_WAVE_BMOV
            // four 64-bit moves under the lanes' mask instead of ten 32-bit selects (the compiler prefers the selects)
            {
                const unsigned long long cm = __ballot(c);
                unsigned long long sv;
                asm volatile("s_and_saveexec_b64 %[sv], %[cm]\n\tv_mov_b64 %[ss], %[s]\n\tv_mov_b64 %[sa], %[fa]\n\t"
                             "v_mov_b64 %[fa], 0\n\tv_mov_b64 %[e], %[en]\n\ts_mov_b64 exec, %[sv]"
                             : [ss] "+v"(snapS), [sa] "+v"(snapA), [fa] "+v"(favg), [e] "+v"(erl), [sv] "=&s"(sv)
                             : [s] "v"(S), [en] "v"(erl_n), [cm] "s"(cm)
                             : "scc");
            }
#else
            snapS = c ? S : snapS;
            snapA = c ? favg : snapA;

            favg = c ? 0.0 : favg;
            erl = c ? erl_n : erl;
#endif
            if (FOLD) {      // the folded leaf crosses the month with its carrier
                snapSL = c ? SL : snapSL;
                snapAL = c ? favgL : snapAL;
                favgL = c ? 0.0 : favgL;
                erlL = c ? erlL_n : erlL;
            }
            if (!MID) {
                if (edge == 1) S = c ? S0v : S;
                if (FOLD && edge == 1) SL = c ? S0L : SL;
                if (edge == 2) {
                    if (c) fend[lane] = F;
                    if (FOLD && c) fend[LANES + lane] = FL;
                }
            }
        }
        // values for the NEXT sub-step: produced during the previous iteration.  The scheduling barriers keep the reads
        // here, a whole sub-step ahead of the sums that consume them.
        __builtin_amdgcn_sched_barrier(0);
        val_t(&ac)[PRE_N] = va[j & 1], (&bc)[POST_N] = vb[j & 1], (&an)[PRE_N] = va[(j & 1) ^ 1], (&bn)[POST_N] = vb[(j & 1) ^ 1];
        const val_t rc = vr[j & 1];
        const unsigned so = (unsigned)((j + RING - 1) & (RING - 1)) * SLOTB;
        if (CHAIN) vr[(j & 1) ^ 1] = *(lds_cv *)(eprv + so);
#pragma unroll
        for (int w = 0; w < PRE; ++w) an[w] = *(lds_cv *)(epre[w] + so);
#pragma unroll
        for (int w = 0; w < POST; ++w) bn[w] = *(lds_cv *)(epost[w] + so);
        __builtin_amdgcn_sched_barrier(0);
        // everything older than the reads just issued and the value stored at the end of the previous sub-step has
        // returned (LDS answers in order; the loop holds no scalar loads): one counted wait per sub-step
        __builtin_amdgcn_s_waitcnt(0xC07F | ((PRE + POST + (CHAIN ? 1 : 0) + 1) << 8));
        __builtin_amdgcn_sched_barrier(0);
        if ((j & (RING - 1)) == 0) block_io(j / RING);
        if ((j & (RING - 1)) == 1) block_store(n + j - 1);
        const double F0 = S * tauinv;                                          // mrtm.py:50
        if constexpr (RSUM) {
            double base = __builtin_fma(S, acoef, erl);                       // (erl: lateral inflow x dt in this form)
            if (FOLD) {      // the folded leaf's step: its outflow joins the cell's inflow of this very sub-step
                FL = SL * tauL;                                                // mrtm.py:50
                SL = __builtin_fma(SL, acoefL, erlL);                          // mrtm.py:51, 69 for a cell without inflow that cannot fire
                favgL += FL;                                                   // mrtm.py:78
                base = __builtin_fma(FL, dt, base);
            }
            if constexpr (SGL == 1) {      // single unit (the template's comment): one running sum, 8-byte entries
                const double S1 = PRE ? __builtin_fma(ac[0], dt, base) : base;
                const double m = __builtin_fmin(S1, 0.0);
                const double f2 = __builtin_fma(m, dtinv, F0);                 // mrtm.py:60
                own[(j & (RING - 1)) * NSLOT] = CHAIN ? rc + f2 : f2;
                S = S1 - m;                                                    // mrtm.py:63, 69
                F = f2;
                favg += f2;                                                    // mrtm.py:78
            } else {
            const double S1 = PRE ? __builtin_fma(ac[0].x, dt, base) : base;  // trial storage: S + dSdt dt (mrtm.py:51, 54)
            const double S2 = PRE ? __builtin_fma(ac[0].y, dt, base) : base;  // the same with the adjusted inflows (mrtm.py:66-69)
            const bool sx = S1 < 0.0;                                          // mrtm.py:54: dSdt dt < -S
            // mrtm.py:60 as F + min(S1, 0) / dt: v_min_f64 + v_fma_f64 instead of the fma and two selects (min(NaN, 0) = 0
            // leaves F2 = F = NaN for a cell whose storage is NaN, as the reference does)
            const double f2 = __builtin_fma(__builtin_fmin(S1, 0.0), dtinv, F0);
            own[(j & (RING - 1)) * NSLOT] = CHAIN ? v2d{rc.x + F0, rc.y + f2} : v2d{F0, f2};
            double Sn = S2;
            asm volatile("" : "+v"(Sn));
            S = sx ? 0.0 : Sn;                                                 // mrtm.py:63, 69
            F = f2;
            favg += f2;                                                        // mrtm.py:78
            if (SGL == 2) fmin_seen = __builtin_fmin(fmin_seen, f2);           // exit guard (looked at by check())
            }
        } else {
            double s1 = 0.0, s2 = 0.0;                                         // UM.dot(F), stored order (mrtm.py:51)
#pragma unroll
            for (int w = 0; w < PRE; ++w) {
                s1 += ac[w].x;
                s2 += ac[w].y;
            }
            s1 -= F0;
#pragma unroll
            for (int w = 0; w < POST; ++w) s1 += bc[w].x;
            const double dsdt = s1 + erl;
            const bool sx = (dsdt * dt) < (-S);                                // mrtm.py:54
            const double f2 = sx ? (dsdt + F0) + S * dtinv : F0;               // mrtm.py:60
            own[(j & (RING - 1)) * NSLOT] = CHAIN ? v2d{rc.x + F0, rc.y + f2} : v2d{F0, f2};
            // second sum with the adjusted flows (mrtm.py:66-69); equal to the first, bit for bit, when nothing it
            // gathers was adjusted, which is the reference's "no cell fired" branch (mrtm.py:76)
            s2 -= f2;
#pragma unroll
            for (int w = 0; w < POST; ++w) s2 += bc[w].y;
            const double dsdt2 = s2 + erl;
            double Sn = S + dsdt2 * dt;
            asm volatile("" : "+v"(Sn));            // keeps the second sum out of an exec-masked region
            S = sx ? 0.0 : Sn;                                                 // mrtm.py:63, 69
            F = f2;
            favg += f2;                                                        // mrtm.py:78
        }
    };

#ifdef XH_WAVE_PROFILE      // diagnostic build (make PROFILE=1): where a unit's cycles go; st[0] / st[4] / st[5] change meaning
    unsigned long long prof_zone = 0, prof_fin = 0, prof_t = __builtin_amdgcn_s_memtime();
#define PROF_MARK(acc)                                                  \
    {                                                                   \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
        acc += now_ - prof_t;                                           \
        prof_t = now_;                                                  \
    }
    unsigned long long prof_plain = 0;
#else
#define PROF_MARK(acc)
#endif
    // Two inner loops -- runs of ordinary groups and runs of boundary groups -- instead of one loop that picks a variant
    // per group: a loop with two unrolled bodies left the values in flight (gathered pairs, import blocks) in different
    // registers on its two paths and the compiler reconciled them at the back-edge with ~30 moves behind
    // s_waitcnt vmcnt(0), draining every import load a group had just issued.
    auto housekeeping = [&](int n) {
        if (n > 0 && (n & (CH - 1)) == 0) check(n);
        if (alive && n == nf) {
            const int g_next = finalize(itf);
            ++itf;
            nf = itf <= nit ? ((g_next + lmax + 1 + GROUP - 1) & ~(GROUP - 1)) : INT_MAX;
        }
        PROF_MARK(prof_fin)
    };
    int n = 0;
    while (n < N && alive) {
        if (itz <= nit && n + GROUP > gz && n <= gz + lmax) {      // boundary groups of month itz
            const int edge = itz == 0 ? 1 : (itz == nit ? 2 : 0);
            auto zone_run = [&](auto mid_c) {
                do {
                    housekeeping(n);
                    if (!alive) break;
                    ++zone_groups;
                    const int rel = nx - n;      // the sub-step of this group at which the lane crosses (outside 0..15: not in this group)
#pragma unroll
                    for (int j = 0; j < GROUP; ++j) substep(std::true_type(), mid_c, n, j, edge, rel);
                    n += GROUP;
                    PROF_MARK(prof_zone)
                } while (n <= gz + lmax && n < N);
            };
#if XH_WAVE_MIDZONE
            if (edge == 0 && !odd_ok) zone_run(std::true_type());
            else zone_run(std::false_type());
#else
            zone_run(std::false_type());
#endif
            if (alive) {      // every lane has crossed: next boundary
                nx = itz < nit ? nx + ntz : INT_MAX;
                ++itz;
                gz = p_rec[itz <= nit ? itz : nit].g;
                if (itz > nit) gz = INT_MAX;
                ntz = p_rec[itz <= nit ? itz : nit].nt;
            }
        } else {                                                   // ordinary groups up to the next boundary
            const int n_end = itz <= nit ? min(N, gz & ~(GROUP - 1)) : N;
            do {
                housekeeping(n);
                if (!alive) break;
#pragma unroll
                for (int j = 0; j < GROUP; ++j) substep(std::false_type(), std::false_type(), n, j, 0, 0);
                n += GROUP;
                PROF_MARK(prof_plain)
            } while (n < n_end);
        }
    }
    if (alive) {
        while (itf <= nit) finalize(itf++);
        if (SGL == 2) gsgl |= (xlane && fmin_seen < -SGL_XEPS) ? 4u : 0u;
        if ((FOLD && __any(gfold != 0)) || (SGL != 0 && __any(gsgl != 0))) {
            __hip_atomic_store(A(fault), FAULT_GUARD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            alive = false;
        }
    }
    if (alive) {
        if (any_x) {      // last block, then everything is published
            const unsigned xpos = ((unsigned)(N - RING - lmax) * 16u) & maskb;
            store_pair(*xsrc[0], xbyte[0], xpos);
            if (x2) store_pair(*xsrc[1], xbyte[1], xpos);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // write-through stores acknowledged
            if (A(fenced) == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (has_x) __hip_atomic_store(A(ready) + xedge, (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (any_g && lane == 0)
            __hip_atomic_store(A(done) + unit, (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (valid) {
            if (A(S_end)) A(S_end)[gc] = snapS;
            if (A(F_end)) A(F_end)[gc] = fend[lane];
        }
        if (validL) {
            if (A(S_end)) A(S_end)[gcL] = snapSL;
            if (A(F_end)) A(F_end)[gcL] = fend[LANES + lane];
        }
    }
    const bool guard_set = __any((gfold | gsgl) != 0);
    if (A(stats) && lane == 0) {
        unsigned long long *st = A(stats) + (int64_t)unit * 6;
        const unsigned long long cyc = __builtin_amdgcn_s_memtime() - cyc_begin;
        st[0] = cyc - cyc_wait_data - cyc_wait_ring;
        st[1] = cyc;
        st[2] = __builtin_amdgcn_s_memrealtime() - rt_begin;
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
        st[3] = (unsigned long long)((PRE + POST + (CHAIN ? 1 : 0) + 1) & 15) | (any_g ? 16u : 0u) | (any_x ? 32u : 0u) |
                (V8 ? 64u : 0u) | (guard_set ? 128u : 0u) | ((unsigned long long)hw << 8) | ((unsigned long long)(xcc & 15u) << 40) |
                (zone_groups << 44);
        st[4] = cyc_wait_data;
        st[5] = cyc_wait_ring;
#ifdef XH_WAVE_PROFILE
        st[0] = prof_plain;      // cycles in groups without a month boundary
        st[4] = prof_zone;       // cycles in boundary groups (their number: st[3] >> 44)
        st[5] = prof_fin;        // checks (with their waits) and month bookkeeping
#endif
    }
}


// ---- which unit this workgroup runs.  The launch has more workgroups than units.  Every workgroup registers on its
//      SIMD and waits until all have (they are all resident: the launch made sure).  First arrivals run a unit; as
//      many second arrivals as there are units left over also do, the rest leave -- so exactly (units - SIMDs in use)
//      SIMDs hold two units however the dispatcher spread the workgroups.  The second arrivals that stay take the
//      cheapest units of the list (units without streams: they delay nobody), their SIMD partners the next ones, with
//      issue priority, everybody else the rest in list order.
//      Single-sum plans (n_excl > 0): the last n_excl units of the list are the PAIR units, which issue 13 fp64 operations
//      per sub-step where their neighbours issue 8 -- among three such neighbours on a CU a pair unit is the slowest unit
//      of the launch.  The first workgroup to register on a CU is its leader; the first leaders (by ticket)
//      keep their CU for their pair unit, its other arrivals leave, and that many more second arrivals elsewhere run
//      a unit.
// Returns the unit (or -1: a spare workgroup, or a fault) and leaves the issue priority of the workgroup in *prio_sh_p.
__device__ __forceinline__ int wave_claim(WaveArgsK *ap, int *unit_sh_p, int *prio_sh_p) {
    int &unit_sh = *unit_sh_p, &prio_sh = *prio_sh_p;
    if (threadIdx.x == 0) {
        prio_sh = 0;
        unsigned *pl = A(place);
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 15u;
        const unsigned key = ((((xcc * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u + ((hw >> 8) & 15u)) * 4u) +
                             ((hw >> 4) & 3u);
        unsigned *pl_cu = pl + 16 + PLACE_KEYS + (key >> 2);
        const int n_units = A(n_units);
        const unsigned n_wg = gridDim.x;
        unsigned *fault = A(fault);
        auto wait_for = [&](unsigned *word, unsigned target) {      // bounded; false and the fault word raised on timeout
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (ld_relaxed(fault) != 0 || __builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT_TICKS) {
                    __hip_atomic_store(fault, FAULT_PLACE_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return false;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            return true;
        };
        auto add = [&](int word) { return __hip_atomic_fetch_add(pl + word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        // words: 0 registered, 1 second arrivals, 2 first arrivals, 3 tickets of the second arrivals, 4 second arrivals
        // decided, 5 / 6 claims of the partners / of everybody else, 7 tickets of the first arrivals, 9 CU leaders, 10 their
        // tickets, 11 first arrivals displaced from exclusive CUs, 12 leaders decided, 13 pair units claimed by exclusive CUs
        const unsigned rank = __hip_atomic_fetch_add(pl + 16 + key, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffu;
        bool leader = false;
        unsigned cu_rank = 0;                          // order of this first arrival among the first arrivals of its CU
        if (rank == 0) {
            add(2);
            cu_rank = __hip_atomic_fetch_add(pl_cu, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffu;
            leader = cu_rank == 0u;
            if (leader) add(9);
        } else if (rank == 1) {
            add(1);
        }
        const unsigned registered = __hip_atomic_fetch_add(pl + 0, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        // fed run: the last workgroup to register tells the host's side stream that every unit is resident and placed --
        // only then may the kernels that produce the rest of the runoff take the free wave slots (xh_fused.hip)
        if (registered + 1u == n_wg && A(place_epoch))
            __hip_atomic_store(A(place_epoch), A(epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int idx = -1;                                  // -1: fault, -2: spare workgroup, nothing to do
        if (wait_for(pl + 0, n_wg)) {
            const int firsts = (int)ld_relaxed(pl + 2), seconds = (int)ld_relaxed(pl + 1), leaders = (int)ld_relaxed(pl + 9);
            // exclusive CUs: ONE pair unit each (two on a CU slow each other down by 15-20 cycles per sub-step: 13.8-14.4 ms
            // against 13.1, round 6), as many as asked for, as long as the second arrivals elsewhere can take over the units of
            // the arrivals that do not run one (up to three first and four second arrivals per CU).  A leader with a ticket
            // claims a pair unit from the END of the list (pl[13]: units claimed so far) and marks its CU.
            // (Sealing the exclusive CUs against other kernels -- their other arrivals resident, asleep -- was built and measured
            // in round 6 and removed: the fillers of a fed step do not cost the pair units anything there;
            // profiles/round6/fed_penalty.txt.)
            const int n_excl = A(n_excl);
            int ncu = min(n_excl, leaders);
            ncu = max(min(ncu, (seconds - max(n_units - firsts, 0)) / 7), 0);
            bool excl_cu = false;
            int displaced = 0, claimed = 0;
            bool ok = true;
            if (ncu > 0) {
                if (leader) {
                    if ((int)add(10) < ncu) {
                        const int cnt = (int)(ld_relaxed(pl_cu) & 0xffffu);
                        const int base = (int)add(13);
                        if (base < n_excl) {
                            __hip_atomic_fetch_or(pl_cu, 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_fetch_add(pl + 11, (unsigned)(cnt - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            idx = n_units - 1 - base;
                        }
                    }
                    __hip_atomic_fetch_add(pl + 12, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                ok = wait_for(pl + 12, (unsigned)leaders);
                excl_cu = (ld_relaxed(pl_cu) & 0x10000u) != 0;
                displaced = (int)ld_relaxed(pl + 11);
                claimed = min((int)ld_relaxed(pl + 13), n_excl);
            }
            const int nx = claimed;
            const int firsts_eff = firsts - displaced;
            const int need2 = max(n_units - firsts_eff, 0);         // second arrivals that must run a unit
            if (idx >= 0 || !ok) {
                // (a pair unit on an exclusive CU, or a fault)
            } else if (rank >= 2) {
                idx = -2;
            } else if (rank == 1) {
                const int t = excl_cu ? INT_MAX : (int)add(3);
                if (t < need2) __hip_atomic_fetch_or(pl + 16 + key, 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(pl + 4, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                idx = t < need2 ? t : -2;
                if (t < need2) prio_sh = 1;
            } else if (excl_cu) {
                idx = -2;                              // the CU belongs to its leader's pair unit
            } else if (wait_for(pl + 4, (unsigned)seconds)) {
                const bool shared = (ld_relaxed(pl + 16 + key) & 0x10000u) != 0;
                if (firsts_eff > n_units && (int)add(7) >= n_units - nx) {
                    idx = -2;
                } else if (shared) {
                    idx = need2 + (int)add(5);
                    prio_sh = 3;
                } else {
                    idx = 2 * need2 + (int)add(6);
                }
                if (idx >= n_units - nx) idx = -1;      // cannot happen: the ranges add up to the units
            }
            if (idx >= n_units) idx = -1;
        }
        if (idx == -1) __hip_atomic_store(fault, FAULT_PLACE_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unit_sh = idx >= 0 ? A(unit_order)[idx] : -1;
    }
    __syncthreads();
    return unit_sh;
}

}  // namespace
