// MRTM routing, time-skewed inside each single-wave unit (gfx950).
//
// xh_mrtm_flow.hip removed the workgroup barriers from the sub-step but kept its two LDS round trips: write the flows,
// read the neighbours' back, sum, and -- when a cell fired (mrtm.py:56-76) -- write the adjusted flows and read them
// again.  A lone wave waits out every one of those trips, and a unit with a cell that fires most sub-steps (channel
// shorter than velocity x dt) pays both, ~750 cycles per sub-step; the slowest unit paces the whole chip.
//
// The dependency only points downstream, so the lanes of a unit need not be at the same sub-step.  Here a cell `h`
// edges above its piece's outlet runs 2 (H - h) sub-steps behind the unit's clock: at iteration n it integrates
// sub-step n - lag.  Everything a cell gathers at iteration n -- the trial flow F AND the adjusted flow F2 of each
// upstream neighbour, for that same sub-step -- was then written to LDS two iterations earlier, as one 16-byte pair:
//   - LDS holds a ring of 8 slots (iteration mod 8) of {F, F2} pairs: 64 cells, 64 imported streams ("ghosts"), one
//     zero;
//   - the reads for iteration n + 1 are issued at the top of iteration n, a whole iteration ahead of their use, so no
//     LDS latency sits on the critical path, and both sums of mrtm.py (first with F, second with F2) come from the
//     same registers: the "fired" path costs one more chain of adds instead of a second round trip;
//   - the own term -F (the diagonal of UM = UP - I) is taken from registers; the row is summed in its stored order as
//     [terms before the diagonal] - F [terms after], each side padded with +0.0 (an accumulator that started at +0.0
//     is never -0.0, so the padding cannot change a bit).  Every operation and its order match scipy's CSR mat-vec,
//     so results stay bit-identical to numpy/scipy.
//   - CHAIN specialisations: the cells that feed a cell from in front of its diagonal pass a running sum along their
//     stored order instead of being read one by one (see skew_unit); the additions are the row sum's own.
// What remains per sub-step is the recurrence itself (~15 dependent fp64 operations) and ~35 instructions of issue
// for the median unit (5 pairs read).
//
// Which workgroup runs which unit is settled at the top of the kernel, not by the workgroup id: every workgroup finds
// the SIMD it landed on and claims a unit so that only units without streams ever share a SIMD (k_mrtm_skew).
//
// Streams between units are rings of RS sub-steps indexed by the global sub-step (not by month) and move in blocks
// of 8 sub-steps, once per 8 iterations, by the whole wave: lane (k, i) stores sub-step i of the unit's k-th outlet
// from the LDS ring (128 contiguous bytes per outlet), and loads sub-step i of its k-th import two blocks ahead of
// dropping it into the ghost entries of the ring.  Producers publish and consumers acquire every CH iterations
// (counters in sub-steps).  Months only matter to a lane when it crosses one: the iterations [G, G + lmax] after the
// unit's clock passes a month start G run a variant of the loop in which the lanes whose lag puts them on the
// boundary snapshot their month-end state and pick up the next month's lateral inflow; the month's outputs are
// formed once, for all lanes, after the last lane has crossed.
#include <algorithm>
#include <climits>
#include <cstdlib>

#include "xh_mrtm_flow.h"

namespace {

constexpr int LANES = 64;
constexpr int SK_P = 4;                       // row terms either side of the diagonal (D8: 4 smaller, 4 larger ids)
constexpr int NSLOT = 2 * LANES + 1;          // pairs per LDS slot: cells, ghosts, constant zero
constexpr unsigned SLOTB = NSLOT * 16u;
constexpr int RING = 8;                       // LDS slots = sub-steps per stream block
constexpr int GROUP = 16;                     // sub-steps per unrolled group (two blocks)
constexpr int SK_R = 2;                       // block-transfer rounds: up to 8 * SK_R imports / outlets per unit
constexpr int CH = 128;                       // iterations between flow-control checks (multiple of GROUP)
constexpr int PUBLAG = 64;                    // a check publishes the stores older than this many iterations
constexpr unsigned FAULT_DATA_WAIT = 1, FAULT_RING_WAIT = 2;
constexpr unsigned FAULT_TEST = 99;
constexpr unsigned FAULT_PLACE_WAIT = 3;
constexpr int PLACE_KEYS = 16 * 8 * 2 * 16 * 4;      // (xcc, se, sh, cu, simd) of HW_ID / XCC_ID
constexpr int PLACE_WORDS = 16 + PLACE_KEYS;

struct SkewArgs {
    const int *cell_of_slot, *lag, *ghost_lag, *export_edge, *ghost_edge, *edge_cons_unit;
    const unsigned *ent2;             // [2][SK_P][units*64] LDS byte offsets inside a slot (before / after the diagonal)
    const unsigned *eprev;            // [units*64] chained units: offset of the pair this lane's own flows are added to
    const int *unit_p, *unit_lmax, *unit_glmax;
    int64_t total_slots;
    int nmonths, nit, total;
    const int *unit_order;            // [units] claim list: units without streams by rising cost, then the others
    unsigned *place;                  // [PLACE_WORDS] counters of claim_unit, zeroed before every launch
    int n_units;
    int odd_ok;                       // 0: months have an even number of sub-steps, so lanes only cross a month start at even iterations
    const int *sched_m, *sched_nt, *sched_g;
    const double *sched_secs;
    const unsigned char *sched_write;
    double dt, dtinv;
    const double *flow_dist, *velocity, *area, *runoff, *S0;
    double *chs, *avg, *S_end, *F_end;
    char *xbuf;                       // [edges][RS] {F, F2}
    unsigned xbytes;                  // size of the rings
    unsigned ring_mask_b;             // RS * 16 - 1
    int rs;                           // RS
    unsigned *ready;                  // [edges] sub-steps published
    unsigned *done;                   // [units] sub-steps consumed
    unsigned *fault;
    unsigned long long *stats;
    unsigned *trace;                  // [units][nit + 1] 100 MHz ticks at which each unit finished each month (XH_FLOW_TRACE, with stats)
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Bound of one wait: 5 s of the 100 MHz real-time counter.  A unit waits for another only while that one is behind,
// i.e. at most about the run time of the whole kernel (35 ms for 67,420 cells x 720 months), so this is two orders of
// magnitude of slack; when it does expire (units of two dataflow kernels sharing the device) the call is re-routed
// with one workgroup per network (xh_fault_check), so a wrong guess costs time, not results.  It is a literal on
// purpose: passing the bound in (kernel argument, fault block, or derived from the sub-step count) kept one more
// scalar live through the sub-step loop and cost 2-5 % of the kernel in spills (measured: 32.3 / 33.0 / 35.1 ms).
constexpr unsigned long long SPIN_LIMIT_TICKS = 500000000ull;

// Lanes with `need` wait until *p >= target (per lane); `seen` keeps the last value each lane read, so that the next
// check can skip the poll (a counter only grows; one poll is an agent-coherent load, ~1-2 us).  False (and the fault
// word raised) on timeout / fault.
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
typedef __attribute__((address_space(3))) const v2d lds_cd2;
typedef __attribute__((address_space(3))) v2d lds_d2;

// The arguments live in device memory and every use re-reads the field it needs (scalar loads of a laundered pointer): held
// as kernel arguments, the ~70 scalar registers of pointers and sizes stayed live across the sub-step loop and were spilled
// to vector lanes and restored around every group of 16 sub-steps.
typedef __attribute__((address_space(4))) const SkewArgs SkewArgsK;      // constant address space: always scalar loads
template <class T>
__device__ __forceinline__ T xh_ldarg(__attribute__((address_space(4))) const T *p) {
    asm volatile("" : "+s"(p));
    return *p;
}
#define A(f) xh_ldarg(&ap->f)

template <int PRE, int POST, bool HAS_G, bool CHAIN>
__device__ __forceinline__ void skew_unit(SkewArgsK *ap, v2d *lds, uint2 *xtab, const int unit) {
    lds_cchar *lds0 = (lds_cchar *)lds;
    const int lane = threadIdx.x;
    const int64_t slot = (int64_t)unit * LANES + lane;

    const int gc = A(cell_of_slot)[slot];
    const bool valid = gc >= 0;
    const double tauinv = valid ? A(velocity)[gc] / A(flow_dist)[gc] : 0.0;      // mrtm.py:40
    const double area = valid ? A(area)[gc] : 0.0;
    const double S0v = (valid && A(S0)) ? A(S0)[gc] : 0.0;
    lds_cchar *epre[PRE], *epost[POST];
#pragma unroll
    for (int w = 0; w < PRE; ++w) epre[w] = lds0 + A(ent2)[(int64_t)w * A(total_slots) + slot];
#pragma unroll
    for (int w = 0; w < POST; ++w) epost[w] = lds0 + A(ent2)[(int64_t)(SK_P + w) * A(total_slots) + slot];
    // CHAIN: the terms in front of a row's diagonal are summed on the way.  The cells that feed one cell from in front of
    // its diagonal form a chain in their stored order; each runs one level (two iterations) behind the one before it and
    // stores {running sum + F, running sum + F2} instead of {F, F2}: the running pair of the cell before it (eprv; the
    // constant zero for the first) plus its own flows.  The cell they feed reads the last one's pair as ONE term.  The
    // additions and their order are those of the row sum ((0 + F_1) + F_2) + ...: same bits, PRE - 1 reads and
    // 2 (PRE - 2) adds fewer per sub-step for a unit whose longest front side is PRE >= 3.
    lds_cchar *eprv = lds0 + (CHAIN ? A(eprev)[slot] : 0u);
    lds_d2 *own = (lds_d2 *)lds + lane;
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
    // The streams are read and written through a buffer resource with the agent-coherent cache policy: stores write
    // through to memory (whole 128-byte lines per outlet and block), loads re-fetch.  No release / acquire fence is
    // needed around the counters -- an agent-scope release writes back the whole L2 of the XCD and was measured to slow
    // every unit on it, importing or not, by ~10 %.
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(A(xbuf), 0, (int)A(xbytes), 0x00020000);
    const int total = A(total), nit = A(nit);
    const bool odd_ok = A(odd_ok) != 0;

    // ---- block-transfer roles: lane (k = lane / 8 + 8 r, i = lane % 8) moves sub-step i of outlet / import k
    if (has_x) xtab[__popcll(xmask & ((1ull << lane) - 1ull))] = make_uint2((unsigned)lane, (unsigned)xedge);
    const int sub = lane & 7, grp = lane >> 3;
    const int nxr = (nx_out + 7) >> 3;
    unsigned gbyte[SK_R], xbyte[SK_R];      // byte offset of the ring (+ 16 i for stores)
    int glag[SK_R];
    bool gon[SK_R], xon[SK_R];
    lds_d2 *gdst[SK_R];
    lds_cd2 *xsrc[SK_R];
#pragma unroll
    for (int r = 0; r < SK_R; ++r) {
        const int k = r * 8 + grp;
        gon[r] = k < ng;
        const int ge = gon[r] ? A(ghost_edge)[(int64_t)unit * LANES + k] : 0;
        glag[r] = gon[r] ? A(ghost_lag)[(int64_t)unit * LANES + k] : 0;
        gbyte[r] = (unsigned)ge * (maskb + 1u);
        gdst[r] = (lds_d2 *)lds + sub * NSLOT + LANES + (k & 63);
        xon[r] = k < nx_out;
        const uint2 t = xon[r] ? xtab[k] : make_uint2(0u, 0u);
        xbyte[r] = t.y * (maskb + 1u) + (unsigned)sub * 16u;
        xsrc[r] = (lds_cd2 *)lds + sub * NSLOT + t.x;
    }

#pragma unroll
    for (int k = 0; k < RING; ++k) {
        own[k * NSLOT] = v2d{0.0, 0.0};
        own[k * NSLOT + LANES] = v2d{0.0, 0.0};
        if (lane == 0) own[k * NSLOT + 2 * LANES] = v2d{0.0, 0.0};
    }

    const double dt = A(dt), dtinv = A(dtinv);
    double S = 0.0, F = 0.0, favg = 0.0, erl = 0.0;
    double snapS = 0.0, snapA = 0.0, snapF = 0.0;
    int nx = A(lag)[slot];                                  // iteration at which this lane enters its next month
    double erl_n = 0.0, qn = 0.0;                          // lateral inflow of the month to enter / runoff after that
    {
        const double q0 = valid ? A(runoff)[(int64_t)gc * A(nmonths) + A(sched_m)[0]] : 0.0;
        erl_n = (q0 * area) * 1000.0 / A(sched_secs)[0];                        // mrtm.py:45
        if (nit > 1) qn = valid ? A(runoff)[(int64_t)gc * A(nmonths) + A(sched_m)[1]] : 0.0;
    }
    // month outputs leave as groups of OB months per cell (32 bytes = one memory sector); 8 months per group held 16
    // more vector registers per lane across the whole sub-step loop
    constexpr int OB = 4;
    double ob_s[OB], ob_a[OB];
#pragma unroll
    for (int j = 0; j < OB; ++j) ob_s[j] = ob_a[j] = 0.0;
    bool alive = true;
    unsigned long long cyc_wait_data = 0, cyc_wait_ring = 0, zone_groups = 0;
    const unsigned long long cyc_begin = __builtin_amdgcn_s_memtime();
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();

    // ---- flow control, every CH iterations, at a group start.  Every block of 8 iterations issues at least one
    //      stream access, so "all but the 8 youngest memory operations have completed" covers every store older than
    //      PUBLAG iterations without draining the loads that are two blocks ahead.
    //      MEMORY-ORDERING ASSUMPTION (outside the HIP memory model, stated here because everything rests on it): the
    //      stream stores are write-through `sc1` buffer stores; vmcnt retires this wave's memory operations in issue
    //      order and a write-through store retires only when the memory side has acknowledged it, so after
    //      `s_waitcnt vmcnt(8)` every store older than PUBLAG iterations is visible at agent scope; only then is the
    //      counter advanced (relaxed agent-scope store, itself ordered behind the waitcnt by the "memory" clobber).  A
    //      consumer reads the counter with an agent-scope load and the data with `sc1` loads, which re-fetch past its
    //      XCD's L2.  No release / acquire fences (an agent-scope release writes back the XCD's whole L2: -10 %).
    //      Evidence: 25 full-size launches per suite run bit-identical to the oracle, two contexts routing concurrently,
    //      the 800-case fuzzer; a violation would show as a wrong bit, and a lost wake-up as a bounded-wait fault.
    //      A poll is an agent-coherent load: 1-2 us during which a lone wave does nothing, i.e. ~30 cycles per sub-step
    //      when every check needs one -- and a unit whose neighbour is only just ahead of it does (measured: every unit
    //      with streams ran at 381-395 cycles per sub-step with no unit's own loop above 364).  So each check asks for
    //      the counters the NEXT check will look at (pend_*, loaded by inline asm so that no wait is attached to them) and
    //      first looks at the values asked for 128 iterations ago; only if those do not cover its needs does it poll and
    //      wait.  Counters only grow, so a stale value is merely conservative.  The `s_waitcnt vmcnt(8)` at the top also
    //      covers the pending counter loads: a unit with streams issues at least 16 memory operations between checks.
    unsigned seen_ready = 0, seen_done = 0, pend_ready = 0, pend_done = 0;
    auto load_async = [&](const unsigned *q) {
        unsigned v;
        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(q) : "memory");
        return v;
    };
    auto check = [&](int n) {
        const unsigned long long w0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(pend_ready), "+v"(pend_done) : : "memory");   // older stores acknowledged, pend_* in
        if (any_x) {      // publish what has certainly been stored, then make sure the next CH iterations have ring space
            const int pub = min(max(n - PUBLAG - RING - lmax, 0), total);
            if (has_x) __hip_atomic_store(A(ready) + xedge, (unsigned)pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int need = n + CH - lmax - A(rs);
            seen_done = max(seen_done, pend_done);
            if (need > 0)
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
            asm volatile("" ::: "memory");      // the stream loads stay behind the poll
        }
        if (any_x) pend_done = load_async(done_p);
        if (any_g) pend_ready = load_async(ready_p);
        cyc_wait_ring += w1 - w0;
        cyc_wait_data += __builtin_amdgcn_s_memtime() - w1;
    };

    // ---- month bookkeeping for all lanes at once: outputs of iteration it - 1, lateral inflow of iteration it + 1
    auto finalize = [&](int it) {
        if (A(trace) && lane == 0) A(trace)[(int64_t)unit * (nit + 1) + it] = (unsigned)(__builtin_amdgcn_s_memrealtime() - rt_begin);
        if (it >= 1) {
            const int m = A(sched_m)[it - 1], ntp = A(sched_nt)[it - 1];
#pragma unroll
            for (int j = 0; j < OB - 1; ++j) {
                ob_s[j] = ob_s[j + 1];
                ob_a[j] = ob_a[j + 1];
            }
            ob_s[OB - 1] = snapS;
            ob_a[OB - 1] = snapA / (double)ntp;                                // mrtm.py:80
            if (A(sched_write)[it - 1] && valid) {     // whole groups of OB months per cell
                const int nmo = A(nmonths);
                if ((m & (OB - 1)) == OB - 1) {
                    const int64_t o = (int64_t)gc * nmo + (m - (OB - 1));
#pragma unroll
                    for (int j = 0; j < OB; j += 2) {
                        if (A(chs)) *reinterpret_cast<v2d *>(A(chs) + o + j) = v2d{ob_s[j], ob_s[j + 1]};
                        if (A(avg)) *reinterpret_cast<v2d *>(A(avg) + o + j) = v2d{ob_a[j], ob_a[j + 1]};
                    }
                } else if (m == nmo - 1) {                                      // last, partial group
                    const int r = (m & (OB - 1)) + 1;
                    const int64_t o = (int64_t)gc * nmo + (m + 1 - r);
#pragma unroll
                    for (int j = 0; j < OB; ++j)
                        if (j >= OB - r) {
                            if (A(chs)) A(chs)[o + j - (OB - r)] = ob_s[j];
                            if (A(avg)) A(avg)[o + j - (OB - r)] = ob_a[j];
                        }
                }
            }
        }
        if (it + 1 < nit) erl_n = (qn * area) * 1000.0 / A(sched_secs)[it + 1];
        if (it + 2 < nit) qn = valid ? A(runoff)[(int64_t)gc * A(nmonths) + A(sched_m)[it + 2]] : 0.0;
    };

    // gathered pairs of the current sub-step (issued one iteration ago); import blocks in flight (two ahead)
    v2d ac[PRE], bc[POST], gbuf[2][SK_R];
    v2d rc = v2d{0.0, 0.0};              // CHAIN: running pair of the cell before this one, current sub-step
#pragma unroll
    for (int w = 0; w < PRE; ++w) ac[w] = v2d{0.0, 0.0};
#pragma unroll
    for (int w = 0; w < POST; ++w) bc[w] = v2d{0.0, 0.0};
    auto import_load = [&](int r, int m0) {      // sub-steps of the block that iterations m0 .. m0 + 7 will drop
        const unsigned pos = ((unsigned)(m0 + sub - glag[r]) * 16u) & maskb;
        return __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(xr, gbyte[r] | pos, 0, AUX_SC1));
    };
    // Block work of the sub-steps m0 = 0 (mod 8), after their gather reads: drop the import block of iterations
    // m0 .. m0 + 7 into the ghost entries, load the block two ahead, read the outlets' block of m0 - 8 .. m0 - 1 out of
    // the LDS ring (before the end of this sub-step overwrites slot 0).  The block is stored one sub-step later
    // (block_store), behind that sub-step's counted wait: stored here, the wave sat out the whole LDS latency of the read
    // it had just issued, once per block (measured: a unit with outlets ran ~35 cycles per sub-step behind one without).
    v4u xb[SK_R];
#pragma unroll
    for (int r = 0; r < SK_R; ++r) xb[r] = v4u{0u, 0u, 0u, 0u};
    auto block_io = [&](int m0, const int b) {
        if (HAS_G) {     // both rounds, unconditionally: lanes past the last import read (and ignore) ring 0
#pragma unroll
            for (int r = 0; r < SK_R; ++r) {
                if (gon[r]) *gdst[r] = gbuf[b][r];
                gbuf[b][r] = import_load(r, m0 + GROUP);
            }
        }
#pragma unroll
        for (int r = 0; r < SK_R; ++r)
            if (r < nxr) {
                if (xon[r]) xb[r] = __builtin_bit_cast(v4u, *xsrc[r]);
            }
    };
    auto block_store = [&](int m0) {
        const unsigned xpos = ((unsigned)(m0 - RING - lmax) * 16u) & maskb;   // 8 sub-steps, never wrapping
#pragma unroll
        for (int r = 0; r < SK_R; ++r)
            if (r < nxr) {
                if (xon[r]) __builtin_amdgcn_raw_buffer_store_b128(xb[r], xr, xbyte[r], xpos, AUX_SC1);
            }
    };

    check(0);
#pragma unroll
    for (int r = 0; r < SK_R; ++r) {
        gbuf[0][r] = HAS_G ? import_load(r, 0) : v2d{0.0, 0.0};
        gbuf[1][r] = HAS_G ? import_load(r, RING) : v2d{0.0, 0.0};
    }

    const int N = (total + lmax + 1 + GROUP - 1) & ~(GROUP - 1);
    int itz = 0, gz = 0;                 // month whose start zone [gz, gz + lmax] is next (itz == nit: the end zone)
    int itf = 0, nf = (lmax + 1 + GROUP - 1) & ~(GROUP - 1);     // next month bookkeeping and its iteration
    int ntz = nit > 0 ? A(sched_nt)[0] : 0;

    for (int n = 0; n < N && alive; n += GROUP) {
        if (n > 0 && (n & (CH - 1)) == 0) {
            check(n);
            if (!alive) break;
        }
        if (n == nf) {
            finalize(itf);
            ++itf;
            nf = itf <= nit ? ((A(sched_g)[itf] + lmax + 1 + GROUP - 1) & ~(GROUP - 1)) : INT_MAX;
        }
        const bool zone = itz <= nit && n + GROUP > gz && n <= gz + lmax;

        auto substep = [&](const int j, const bool in_zone) {
            // lanes crossing a month start (or starting / finishing the series) at this iteration.  Lane lags are even and
            // so are the month lengths of any dt that divides 12 h: the odd sub-steps of a group then have no lane to look for
            if (in_zone && ((j & 1) == 0 || odd_ok)) {
                if (nx == n + j) {
                    snapS = S;
                    snapA = favg;
                    snapF = F;
                    favg = 0.0;
                    erl = erl_n;
                    if (itz == 0) S = S0v;
                    nx = itz < nit ? nx + ntz : INT_MAX;
                }
            }
            // pairs for the NEXT sub-step: produced during the previous iteration.  The scheduling barriers keep the
            // reads here, a whole sub-step ahead of the sums that consume them (left alone, the scheduler pulls the next
            // sub-step's first adds up to ~15 instructions behind the reads and the wave waits out the LDS latency).
            __builtin_amdgcn_sched_barrier(0);
            v2d an[PRE], bn[POST], rn = v2d{0.0, 0.0};
            if (CHAIN) rn = *(lds_cd2 *)(eprv + ((j + RING - 1) & (RING - 1)) * SLOTB);
#pragma unroll
            for (int w = 0; w < PRE; ++w) an[w] = *(lds_cd2 *)(epre[w] + ((j + RING - 1) & (RING - 1)) * SLOTB);
#pragma unroll
            for (int w = 0; w < POST; ++w) bn[w] = *(lds_cd2 *)(epost[w] + ((j + RING - 1) & (RING - 1)) * SLOTB);
            __builtin_amdgcn_sched_barrier(0);
            // everything older than the PRE + POST reads just issued and the pair stored at the end of the previous
            // sub-step has returned (LDS answers in order; the loop holds no scalar loads): one counted wait here instead
            // of one in front of every add that consumes a pair.  The store is left out: a unit with few terms would wait
            // for it (measured: 220 -> 254 cycles per sub-step at 2 terms).
            __builtin_amdgcn_s_waitcnt(0xC07F | ((PRE + POST + (CHAIN ? 1 : 0) + 1) << 8));
            __builtin_amdgcn_sched_barrier(0);
            if ((j & (RING - 1)) == 0) block_io(n + j, j / RING);
            if ((j & (RING - 1)) == 1) block_store(n + j - 1);
            const double F0 = S * tauinv;                                      // mrtm.py:50
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
            asm volatile("" : "+v"(Sn));            // keeps the second sum out of an exec-masked region (4 more instructions)
            S = sx ? 0.0 : Sn;                                                 // mrtm.py:63, 69
            F = f2;
            favg += f2;                                                        // mrtm.py:78
            if (CHAIN) rc = rn;
#pragma unroll
            for (int w = 0; w < PRE; ++w) ac[w] = an[w];
#pragma unroll
            for (int w = 0; w < POST; ++w) bc[w] = bn[w];
        };

        if (zone) {
            ++zone_groups;
#pragma unroll
            for (int j = 0; j < GROUP; ++j) substep(j, true);
            if (n + GROUP > gz + lmax) {      // every lane has crossed: next boundary
                ++itz;
                gz = itz <= nit ? A(sched_g)[itz] : INT_MAX;
                ntz = itz < nit ? A(sched_nt)[itz] : 0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < GROUP; ++j) substep(j, false);
        }
    }
    if (alive) {
        while (itf <= nit) finalize(itf++);
        if (any_x) {      // last block, then everything is published
            const unsigned xpos = ((unsigned)(N - RING - lmax) * 16u) & maskb;
#pragma unroll
            for (int r = 0; r < SK_R; ++r)
                if (r < nxr) {
                    if (xon[r])
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, *xsrc[r]), xr, xbyte[r], xpos, AUX_SC1);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // write-through stores acknowledged
            if (has_x) __hip_atomic_store(A(ready) + xedge, (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (any_g && lane == 0)
            __hip_atomic_store(A(done) + unit, (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (valid) {
            if (A(S_end)) A(S_end)[gc] = snapS;
            if (A(F_end)) A(F_end)[gc] = snapF;
        }
    }
    if (A(stats) && lane == 0) {
        unsigned long long *st = A(stats) + (int64_t)unit * 6;
        const unsigned long long cyc = __builtin_amdgcn_s_memtime() - cyc_begin;
        st[0] = cyc - cyc_wait_data - cyc_wait_ring;
        st[1] = cyc;
        st[2] = __builtin_amdgcn_s_memrealtime() - rt_begin;
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
        st[3] = (unsigned long long)(PRE + POST + (CHAIN ? 1 : 0) + 1) | (any_g ? 16u : 0u) | (any_x ? 32u : 0u) |
                ((unsigned long long)hw << 8) | ((unsigned long long)(xcc & 15u) << 40) | (zone_groups << 44);
        st[4] = cyc_wait_data;
        st[5] = cyc_wait_ring;
    }
}

// Two waves per SIMD must fit (1,121 units of the 67,420-cell grid on 1,024 SIMDs): at most 256 registers.
__global__ void __launch_bounds__(LANES) __attribute__((amdgpu_waves_per_eu(2, 2))) k_mrtm_skew(const SkewArgs *ap_) {
    SkewArgsK *ap = (SkewArgsK *)ap_;
    __shared__ __attribute__((aligned(16))) v2d lds[RING * NSLOT];
    __shared__ uint2 xtab[LANES];
    // ---- which unit this workgroup runs (see xh_mrtm_flow.hip, "which unit runs where").  The launch has more workgroups
    //      than units.  Every workgroup registers on its SIMD and waits until all have (they are all resident: the launch
    //      made sure).  First arrivals run a unit; as many second arrivals as there are units left over also do, the rest
    //      leave -- so exactly (units - SIMDs in use) SIMDs hold two units however the dispatcher spread the workgroups
    //      (without the spare workgroups it doubled up 38-65 SIMDs in a pipelined run and left as many empty).  The second
    //      arrivals that stay take the cheapest units of the list, their SIMD partners the next ones, with issue
    //      priority, everybody else the rest in list order.
    __shared__ int unit_sh;
    if (threadIdx.x == 0) {
        unsigned *pl = A(place);
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 15u;
        const unsigned key = ((((xcc * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u + ((hw >> 8) & 15u)) * 4u) +
                             ((hw >> 4) & 3u);
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
        // decided, 5 / 6 claims of the partners / of everybody else, 7 tickets of the first arrivals
        const unsigned rank = __hip_atomic_fetch_add(pl + 16 + key, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffu;
        if (rank == 0) add(2);
        else if (rank == 1) add(1);
        __hip_atomic_fetch_add(pl + 0, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int idx = -1;                                  // -1: fault, -2: spare workgroup, nothing to do
        if (wait_for(pl + 0, n_wg)) {
            const int firsts = (int)ld_relaxed(pl + 2), seconds = (int)ld_relaxed(pl + 1);
            const int need2 = max(n_units - firsts, 0);         // second arrivals that must run a unit
            if (rank >= 2) {
                idx = -2;
            } else if (rank == 1) {
                const int t = (int)add(3);
                if (t < need2) __hip_atomic_fetch_or(pl + 16 + key, 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(pl + 4, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                idx = t < need2 ? t : -2;
            } else if (wait_for(pl + 4, (unsigned)seconds)) {
                const bool shared = (ld_relaxed(pl + 16 + key) & 0x10000u) != 0;
                if (firsts > n_units && (int)add(7) >= n_units) {
                    idx = -2;
                } else if (shared) {
                    idx = need2 + (int)add(5);
                    __builtin_amdgcn_s_setprio(3);
                } else {
                    idx = 2 * need2 + (int)add(6);
                }
            }
            if (idx >= n_units) idx = -1;      // cannot happen: the three ranges add up to the units
        }
        if (idx == -1) __hip_atomic_store(fault, FAULT_PLACE_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unit_sh = idx >= 0 ? A(unit_order)[idx] : -1;
    }
    __syncthreads();
    const int unit = unit_sh;
    if (unit < 0) return;
    const int p = A(unit_p)[unit];                   // uniform per workgroup: terms before | after the diagonal << 4
    const bool g = __any(A(ghost_edge)[(int64_t)unit * LANES + threadIdx.x] >= 0);
    // one specialisation per (terms before, terms after, imports?): an LDS read costs a lone wave ~17 cycles of issue
    // whatever its width (tools/micro/substep_cost.hip), so no unit should read padding it does not need
#define SKEW_CASE(PRE, POST)                                         \
    case (PRE) | ((POST) << 4):                                      \
        if (g) skew_unit<PRE, POST, true, false>(ap, lds, xtab, unit);     \
        else skew_unit<PRE, POST, false, false>(ap, lds, xtab, unit);      \
        break;
#define SKEW_CHAIN(PRE, POST)                                        \
    case (PRE) | ((POST) << 4) | 0x100:                              \
        if (g) skew_unit<PRE, POST, true, true>(ap, lds, xtab, unit);      \
        else skew_unit<PRE, POST, false, true>(ap, lds, xtab, unit);       \
        break;
    switch (p) {
        SKEW_CASE(1, 1) SKEW_CASE(1, 2) SKEW_CASE(1, 3) SKEW_CASE(1, 4)
        SKEW_CASE(2, 1) SKEW_CASE(2, 2) SKEW_CASE(2, 3) SKEW_CASE(2, 4)
        SKEW_CASE(3, 1) SKEW_CASE(3, 2) SKEW_CASE(3, 3) SKEW_CASE(3, 4)
        SKEW_CASE(4, 1) SKEW_CASE(4, 2) SKEW_CASE(4, 3)
        // chained units: front side 3 or 4 summed on the way, 1 or 2 terms left to read (xh_mrtm_flow.hip)
        SKEW_CHAIN(1, 1) SKEW_CHAIN(1, 2) SKEW_CHAIN(1, 3) SKEW_CHAIN(1, 4)
        SKEW_CHAIN(2, 1) SKEW_CHAIN(2, 2) SKEW_CHAIN(2, 3) SKEW_CHAIN(2, 4)
        default: if (g) skew_unit<4, 4, true, false>(ap, lds, xtab, unit); else skew_unit<4, 4, false, false>(ap, lds, xtab, unit);
    }
#undef SKEW_CHAIN
#undef SKEW_CASE
}
#undef A

__global__ void k_mrtm_skew_args(SkewArgs a, SkewArgs *dst) {
    if (threadIdx.x == 0) *dst = a;
}

}  // namespace

int skew_launch(xh_ctx *ctx, FlowPlan *fp, const FlowSched &s, const FlowIO &io, hipStream_t st) {
    if (!fp || fp->n_units == 0) return XH_OK;
    // a lane must have left month it - 1 before the unit's clock reaches month it + 1 (one pending snapshot per lane)
    if (!fp->skew_ok || fp->max_imports > 8 * SK_R || fp->max_exports > 8 * SK_R || s.ntmin < fp->skew_lmax + 2 * GROUP) {
        if (getenv("XH_FLOW_DEBUG"))
            fprintf(stderr, "skew kernel not used: skew_ok %d, imports %d, outlets %d, shortest month %d sub-steps, largest lag %d\n",
                    (int)fp->skew_ok, fp->max_imports, fp->max_exports, s.ntmin, fp->skew_lmax);
        return XH_ERR_LIMIT;
    }
    // ring: a consumer asks for ~2 CH + lag sub-steps ahead of its clock, a producer may run RS - CH - lag ahead
    // ring: a consumer asks for ~2 CH + lag sub-steps ahead of its clock, a producer may run RS - CH - lag ahead.  A stream
    // that jumps over k pipeline levels (a tributary that joins the main stem far downstream: its consumer also waits for
    // units k levels below the producer) needs the lead of all of them in its ring: every level trails the one above by
    // PUBLAG + RING + lag + CH + GROUP + up to CH of check granularity ~ 400-450 sub-steps.  With a ring shorter than that
    // nobody deadlocks, but the producer is held at the ring limit, its other consumers starve, and every linked unit ends
    // up waiting a quarter of the time (measured: 32.7 instead of 26.7 ms with 2,048 sub-steps and a 6-level jump).
    int rs = 2048;
    while (rs < 8 * CH + 4 * fp->skew_lmax || rs < 1024 + 512 * fp->skew_span) rs *= 2;
    if (const char *env = getenv("XH_FLOW_RS")) {      // experiments: a power of two
        const int v = atoi(env);
        if (v >= 2048 && (v & (v - 1)) == 0) rs = v;
    }
    const size_t x_streams = (size_t)std::max(fp->n_edges, 1) * (size_t)rs * sizeof(v2d);
    if (x_streams >= ((size_t)1 << 32)) return XH_ERR_LIMIT;      // 32-bit ring offsets
    const size_t x_cnt = ((size_t)(fp->n_edges + fp->n_units + PLACE_WORDS) * sizeof(unsigned) + 255) & ~size_t(255);
    fp->rec_key = 0;      // (k_mrtm_wave keeps its month records in this buffer too: this launch lays it out its own way)
    if (x_streams + x_cnt > fp->x_bytes) {
        if (fp->d_x) {
            XH_HIP(ctx, hipStreamSynchronize(st));
            XH_HIP(ctx, hipFree(fp->d_x));
            fp->d_x = nullptr;
        }
        XH_HIP(ctx, hipMalloc(&fp->d_x, x_streams + x_cnt));
        fp->x_bytes = x_streams + x_cnt;
    }
    unsigned *cnt = reinterpret_cast<unsigned *>(static_cast<char *>(fp->d_x) + x_streams);
    XH_HIP(ctx, hipMemsetAsync(cnt, 0, x_cnt, st));

    // every unit resident at once: see flow_launch for the LDS-share sizing (one workgroup more than the even split)
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    // spare workgroups (k_mrtm_skew, "which unit this workgroup runs"): one per CU, as long as two waves per SIMD hold all
    int n_wg = fp->n_units + cus;
    if (n_wg > 8 * cus) n_wg = std::max(fp->n_units, 8 * cus);
    {
        const char *env = getenv("XH_FLOW_SPARE");            // experiments only
        if (env) n_wg = fp->n_units + std::max(atoi(env), 0);
    }
    int per_cu = (n_wg + cus - 1) / cus + 1;
    {
        const char *env = getenv("XH_FLOW_PER_CU_EXTRA");     // experiments only
        if (env) per_cu += atoi(env);
    }
    const size_t lds_static = (size_t)RING * NSLOT * sizeof(v2d) + LANES * sizeof(uint2) + 64;      // + unit_sh, padded
    const size_t share = ((size_t)(160 * 1024) / (size_t)per_cu) & ~size_t(1023);
    size_t lds = share > lds_static + 1024 ? share - lds_static : 0;
    XH_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mrtm_skew), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    int resident = 0;
    XH_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, k_mrtm_skew, LANES, lds));
    if ((int64_t)(resident - 1) * cus < n_wg || n_wg > 8 * cus) return XH_ERR_LIMIT;

    SkewArgs a;
    a.cell_of_slot = static_cast<const int *>(fp->d_cell_of_slot.p);
    a.lag = static_cast<const int *>(fp->d_lag.p);
    a.ghost_lag = static_cast<const int *>(fp->d_ghost_lag.p);
    a.export_edge = static_cast<const int *>(fp->d_export_edge.p);
    a.ghost_edge = static_cast<const int *>(fp->d_ghost_edge.p);
    a.edge_cons_unit = static_cast<const int *>(fp->d_edge_cons_unit.p);
    a.ent2 = static_cast<const unsigned *>(fp->d_ent2.p);
    a.eprev = static_cast<const unsigned *>(fp->d_eprev.p);
    a.unit_p = static_cast<const int *>(fp->d_unit_p.p);
    a.unit_order = static_cast<const int *>(fp->d_unit_order.p);
    a.n_units = fp->n_units;
    a.unit_lmax = static_cast<const int *>(fp->d_unit_lmax.p);
    a.unit_glmax = static_cast<const int *>(fp->d_unit_glmax.p);
    a.total_slots = (int64_t)fp->n_units * LANES;
    a.nmonths = s.nmonths;
    a.nit = s.nit;
    a.total = s.total;
    a.odd_ok = s.nt_even ? 0 : 1;
    a.sched_m = s.d_m;
    a.sched_nt = s.d_nt;
    a.sched_g = s.d_g;
    a.sched_secs = s.d_secs;
    a.sched_write = s.d_wr;
    a.dt = s.dt;
    a.dtinv = 1.0 / s.dt;
    a.flow_dist = io.flow_dist;
    a.velocity = io.velocity;
    a.area = io.area;
    a.runoff = io.runoff;
    a.S0 = io.S0;
    a.chs = io.chs;
    a.avg = io.avg;
    a.S_end = io.S_end;
    a.F_end = io.F_end;
    a.xbuf = static_cast<char *>(fp->d_x);
    a.xbytes = (unsigned)x_streams;
    a.ring_mask_b = (unsigned)rs * 16u - 1u;
    a.rs = rs;
    a.ready = cnt;
    a.done = cnt + fp->n_edges;
    a.place = cnt + fp->n_edges + fp->n_units;
    unsigned *fault = nullptr;
    int rc = xh_fault_word(ctx, &fault);
    if (rc) return rc;
    a.fault = fault;
    // XH_ROUTE_TEST_FAULT raises the fault word before the launch, as a timed-out wait of another unit would: every
    // unit that has to wait gives up and the call is re-routed.
    if (s.test_fault) XH_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(fault), (int)FAULT_TEST, 1, st));
    a.stats = nullptr;
    {
        const char *env = getenv("XH_FLOW_STATS");
        if (env && env[0] == '1') {
            if (!fp->d_stats) XH_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&fp->d_stats), (size_t)fp->n_units * 48));
            a.stats = fp->d_stats;
            if (getenv("XH_FLOW_TRACE")) {
                if (fp->d_trace) (void)hipFree(fp->d_trace);
                fp->trace_words = (size_t)fp->n_units * (size_t)(s.nit + 1);
                XH_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&fp->d_trace), fp->trace_words * 4));
                XH_HIP(ctx, hipMemsetAsync(fp->d_trace, 0, fp->trace_words * 4, st));
            }
        }
    }
    a.trace = fp->d_trace;
    if (!fp->d_skew_args) XH_HIP(ctx, hipMalloc(&fp->d_skew_args, sizeof(SkewArgs)));
    // stream-ordered: the previous launch has finished reading the block before this one rewrites it
    hipLaunchKernelGGL(k_mrtm_skew_args, dim3(1), dim3(64), 0, st, a, static_cast<SkewArgs *>(fp->d_skew_args));
    hipLaunchKernelGGL(k_mrtm_skew, dim3((unsigned)n_wg), dim3(LANES), lds, st,
                       static_cast<const SkewArgs *>(fp->d_skew_args));
    XH_HIP(ctx, hipGetLastError());
    return XH_OK;
}
