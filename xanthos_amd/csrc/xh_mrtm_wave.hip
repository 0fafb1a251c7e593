// MRTM routing, time-skewed inside single-wave units (gfx950) -- round 3 kernel.
//
// Same dataflow design as round 2's k_mrtm_skew (units of 64 lanes = 64 cells of connected pieces of the river trees, a
// cell `h` edges above its piece's outlet running 2 (H - h) sub-steps behind the unit's clock, one-way streams in HBM
// between units, every unit resident at once, mrtm.py:16-82 evaluated per cell in its two-phase form with every sum in
// scipy's CSR order -> bit-identical to numpy/scipy), rebuilt around what round 2's measurements said:
//
//   * A lone wave pays ~5 cycles per instruction of ANY kind and ~15 per ds_read_b128 (tools/micro/substep_plain.hip:
//     a (2,3) pair sub-step is 186 cycles on its own, 191 with four waves per CU) -- but the same sub-step took 290 in
//     k_mrtm_skew.  The ~100 cycles in between were not the sub-step: (a) the month-boundary variant of the loop, taken
//     ~27 % of the time, whose per-lane `if (crossing)` blocks cost 15 instructions per sub-step and broke the register
//     renaming of the gathered pairs (56 v_mov per 8 sub-steps); (b) ~36 instructions of exec-mask juggling and address
//     arithmetic per 8-sub-step stream block; (c) ~25 dependent scalar loads, each with its own s_waitcnt, per month.
//     Here (a) is branch-free (one compare against an inline constant and five selects on the even sub-steps of a
//     boundary group), (b) has no exec masks at all (idle block lanes write unused ghost entries and access the rings
//     out of range, which the buffer resource drops) and one transfer round unless a unit really has more than 8 imports
//     / outlets, (c) reads one 32-byte record per month.
//
//   (Rounds 3-5 also had PLAIN units here -- one 8-byte value per term where no upstream neighbour can fire, with a guard, a
//   learning pass and per-box caches of the cells seen firing: the bit-exact kernel's way to 22.5 ms.  Since round 5 the
//   reassociated kernel k_mrtm_rsum routes by default and this one is the CHECKER and the XH_ROUTE_EXACT option: the
//   adaptive machinery was retired in round 6 -- every unit in pair form, 25 ms at the full grid, nothing to learn.)
//
// Which workgroup runs which unit is settled at the top of the kernel, not by the workgroup id (see k_mrtm_wave).
#include <algorithm>
#include "xh_mrtm_wave_unit.h"

namespace {

// Two waves per SIMD must fit (more units than SIMDs): at most 256 registers.
__global__ void __launch_bounds__(LANES) __attribute__((amdgpu_waves_per_eu(2, 2))) k_mrtm_wave(const WaveArgs *ap_) {
    WaveArgsK *ap = (WaveArgsK *)ap_;
    __shared__ __attribute__((aligned(16))) v2d lds[RING * NSLOT];
    __shared__ uint2 xtab[LANES];
    __shared__ unsigned qstage_sh[2 * LANES];      // runoff of the month after next, low / high words (runoff_fetch)
    // outflow of every lane's last sub-step (F_end): kept here, not in a register selected at every month start.  (Round 4
    // also tried LDS for the month outputs' group of four -- no gain -- and for cell area / initial storage / next month's
    // lateral inflow: 23.2 -> 26.9 ms, their reads break the sub-step's counted s_waitcnt lgkmcnt.)
    __shared__ double fend_sh[LANES];
    __shared__ int unit_sh, prio_sh;
    const int unit = wave_claim(ap, &unit_sh, &prio_sh);
    const int prio = prio_sh;
    if (unit < 0) return;
    // Issue priority: a unit that shares its SIMD with a cheaper unit runs ahead of it (3 against the partner's 0); in a fed
    // run every unit runs ahead of the waves of the kernels that produce its runoff on the same SIMDs (2, partners 3 / 1).
    if (A(months_ready)) {
        if (prio == 3) __builtin_amdgcn_s_setprio(3);
        else if (prio == 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(2);
    } else if (prio == 3) {
        __builtin_amdgcn_s_setprio(3);
    }
    const int p = A(unit_p)[unit];       // uniform per workgroup: terms before | after the diagonal << 4 | chained << 8
    const bool has_ghost = A(ghost_edge)[(int64_t)unit * LANES + threadIdx.x] >= 0;      // lane k: the unit's k-th import
    const bool g = __any(has_ghost), g2 = __any(has_ghost && threadIdx.x >= 8);
    char *l = reinterpret_cast<char *>(lds);
    __attribute__((address_space(3))) unsigned *qst = (__attribute__((address_space(3))) unsigned *)qstage_sh;
    __attribute__((address_space(3))) double *fnd = (__attribute__((address_space(3))) double *)fend_sh;
    fend_sh[threadIdx.x] = 0.0;
    // one specialisation per (terms before, terms after, imports?, chained?): an LDS read costs a lone wave 8-15 cycles of
    // issue, so no unit should read padding it does not need
    // NG1 = true compiles the one-round form too (the shapes most units have; each form is ~45 KB of code and ~10 s of
    // compile time, so the rare shapes send every unit with imports through the two-round form)
#define WAVE_FORM(PRE, POST, CHAINED, NG1)                                                        \
    case (PRE) | ((POST) << 4) | ((CHAINED) ? 0x100 : 0):                                         \
        if (g2 || (g && !(NG1))) wave_unit<PRE, POST, 2, CHAINED>(ap, l, xtab, qst, fnd, unit);        \
        else if (g) wave_unit<PRE, POST, (NG1) ? 1 : 2, CHAINED>(ap, l, xtab, qst, fnd, unit);         \
        else wave_unit<PRE, POST, 0, CHAINED>(ap, l, xtab, qst, fnd, unit);                            \
        break;
#define WAVE_PAIR(PRE, POST, CHAINED) WAVE_FORM(PRE, POST, CHAINED, false)
#define WAVE_PAIR1(PRE, POST, CHAINED) WAVE_FORM(PRE, POST, CHAINED, true)
    switch (p) {
        WAVE_PAIR1(1, 1, false) WAVE_PAIR1(1, 2, false) WAVE_PAIR1(1, 3, false) WAVE_PAIR1(1, 4, false)
        WAVE_PAIR(2, 1, false) WAVE_PAIR1(2, 2, false) WAVE_PAIR1(2, 3, false) WAVE_PAIR1(2, 4, false)
        WAVE_PAIR(3, 1, false) WAVE_PAIR1(3, 2, false) WAVE_PAIR(3, 3, false) WAVE_PAIR(3, 4, false)
        WAVE_PAIR(4, 1, false) WAVE_PAIR(4, 2, false) WAVE_PAIR(4, 3, false) WAVE_PAIR(4, 4, false)
        // chained units: front side 3 or 4 summed on the way, 1 or 2 terms left to read (flow_plan_build)
        WAVE_PAIR(1, 1, true) WAVE_PAIR1(1, 2, true) WAVE_PAIR1(1, 3, true) WAVE_PAIR(1, 4, true)
        WAVE_PAIR(2, 1, true) WAVE_PAIR1(2, 2, true) WAVE_PAIR1(2, 3, true) WAVE_PAIR(2, 4, true)
        default:      // not produced by the plan; a fault rather than wrong results
            if (threadIdx.x == 0) __hip_atomic_store(A(fault), FAULT_PLACE_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#undef WAVE_PAIR1
#undef WAVE_PAIR
#undef WAVE_FORM
}
#undef A

}  // namespace

const void *wave_exact_kernel() { return reinterpret_cast<const void *>(&k_mrtm_wave); }
