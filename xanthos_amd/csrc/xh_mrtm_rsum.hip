// MRTM routing, reassociated ("tolerance") form of the time-skewed dataflow kernel (gfx950) -- the DEFAULT routing kernel
// since round 5 (XH_REASSOC_DEFAULT in xh_mrtm.hip; XH_ROUTE_EXACT / XH_ROUTE_REASSOC=0 / ini `routing_form = exact` select
// the bit-exact k_mrtm_wave instead, which is also the checker behind XH_ROUTE_VALIDATE).
//
// Same machine as k_mrtm_wave (xh_mrtm_wave.hip; the unit's whole run is wave_unit<> of xh_mrtm_wave_unit.h): single-wave units
// of 64 cells, every unit resident at once, lanes time-skewed by their level, one-way streams in HBM between units, month
// records, fed runs.  What it gives up is the ORDER of the row sum of mrtm.py:50-51, and with it the bits: every upstream
// neighbour of a cell passes a running sum along a chain of lanes (and of pieces: xh_flow_rsum.cpp), so a lane reads two
// values per sub-step whatever its row looks like, and the explicit Euler update with the "excess flow" rule
// (mrtm.py:54-69) is fused.  Results equal the reference's to rounding (<= 1e-9 relative against the oracle over the full
// grid and series; the north star's gate is 1e-6) with identical NaN masks -- ChStorage and Avg_ChFlow are NOT bit-identical
// to the reference in this form.
//
// Two kinds of plan (xh_flow_rsum.cpp):
//   pairs        every lane passes {sum F, sum F2} (16-byte entries, 13 fp64 operations per sub-step); needs nothing but the
//                topology -- what an unprepared plan routes on, and where a guard trip falls back to;
//   single sums  (prepared plans: xh_route_plan_prepare knows which cells can fire) every lane passes the ONE sum of the
//                adjusted flows (8-byte entries, 8 operations), except the few cells that may fire AND have an upstream
//                neighbour that may: those sit in pair units of their own, which get a CU to themselves (wave_claim).
// Folded leaves (prepared plans, units without streams) ride in their parents' lanes in either kind.
#include <algorithm>
#include "xh_mrtm_wave_unit.h"

namespace {

__global__ void __launch_bounds__(LANES) __attribute__((amdgpu_waves_per_eu(2, 2))) k_mrtm_rsum(const WaveArgs *ap_) {
    WaveArgsK *ap = (WaveArgsK *)ap_;
    __shared__ __attribute__((aligned(16))) v2d lds[RING * NSLOT];
    __shared__ uint2 xtab[LANES];
    __shared__ unsigned qstage_sh[4 * LANES];      // runoff of the month after next, low / high words (runoff_fetch); second half: folded leaves
    __shared__ double fend_sh[2 * LANES];          // outflow of every lane's last sub-step (F_end); second half: folded leaves
    __shared__ int unit_sh, prio_sh;
    const int unit = wave_claim(ap, &unit_sh, &prio_sh);
    const int prio = prio_sh;
    if (unit < 0) return;
    if (A(months_ready)) {      // fed run: ahead of the waves that produce the runoff on the same SIMDs (see k_mrtm_wave)
        if (prio == 3) __builtin_amdgcn_s_setprio(3);
        else if (prio == 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(2);
    } else if (prio == 3) {
        __builtin_amdgcn_s_setprio(3);
    }
    const int p = A(unit_p)[unit];       // 0x400 | 1 (reads the inflow entry) | 2 (reads the chain entry)
    const bool has_ghost = A(ghost_edge)[(int64_t)unit * LANES + threadIdx.x] >= 0;      // lane k: the unit's k-th import
    const bool g = __any(has_ghost), g2 = __any(has_ghost && threadIdx.x >= 8);
    const bool x = __any(A(export_edge)[(int64_t)unit * LANES + threadIdx.x] >= 0);
    char *l = reinterpret_cast<char *>(lds);
    __attribute__((address_space(3))) unsigned *qst = (__attribute__((address_space(3))) unsigned *)qstage_sh;
    __attribute__((address_space(3))) double *fnd = (__attribute__((address_space(3))) double *)fend_sh;
    fend_sh[threadIdx.x] = 0.0;
    fend_sh[LANES + threadIdx.x] = 0.0;
    if ((p & ~31) != 0x400) {      // not a reassociated plan: a fault rather than wrong results
        if (threadIdx.x == 0) __hip_atomic_store(A(fault), FAULT_PLACE_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    bool bad = false;
    if (p & 16) {             // pair unit of a single-sum plan: the step of the plan of pairs + the exit guard
        if (!(p & 8) || (p & 4)) bad = true;
        else if (g2) wave_unit<1, 0, 2, true, true, false, 2>(ap, l, xtab, qst, fnd, unit);
        else if (g) wave_unit<1, 0, 1, true, true, false, 2>(ap, l, xtab, qst, fnd, unit);
        else wave_unit<1, 0, 0, true, true, false, 2>(ap, l, xtab, qst, fnd, unit);
    } else if (p & 8) {      // single unit: one running sum, 8-byte entries
        if (p & 4) {          // (a unit that carries folded leaves has no imports: xh_flow_rsum.cpp)
            if (g || !A(fold_cell)) bad = true;
            else wave_unit<1, 0, 0, true, true, true, 1>(ap, l, xtab, qst, fnd, unit);
        } else if (g2) wave_unit<1, 0, 2, true, true, false, 1>(ap, l, xtab, qst, fnd, unit);
        else if (g) wave_unit<1, 0, 1, true, true, false, 1>(ap, l, xtab, qst, fnd, unit);
        else if ((p & 2) || x) wave_unit<1, 0, 0, true, true, false, 1>(ap, l, xtab, qst, fnd, unit);
        else if (p & 1) wave_unit<1, 0, 0, false, true, false, 1>(ap, l, xtab, qst, fnd, unit);
        else wave_unit<0, 0, 0, false, true, false, 1>(ap, l, xtab, qst, fnd, unit);
    } else if (p & 4) {
        if (g || !A(fold_cell)) bad = true;
        else wave_unit<1, 0, 0, true, true, true>(ap, l, xtab, qst, fnd, unit);
    } else if (g2) wave_unit<1, 0, 2, true, true>(ap, l, xtab, qst, fnd, unit);
    else if (g) wave_unit<1, 0, 1, true, true>(ap, l, xtab, qst, fnd, unit);
    else if ((p & 2) || x) wave_unit<1, 0, 0, true, true>(ap, l, xtab, qst, fnd, unit);
    else if (p & 1) wave_unit<1, 0, 0, false, true>(ap, l, xtab, qst, fnd, unit);
    else wave_unit<0, 0, 0, false, true>(ap, l, xtab, qst, fnd, unit);
    if (bad && threadIdx.x == 0)      // a shape the planner does not produce: a fault rather than wrong results
        __hip_atomic_store(A(fault), FAULT_PLACE_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#undef A

}  // namespace

const void *wave_rsum_kernel() { return reinterpret_cast<const void *>(&k_mrtm_rsum); }
