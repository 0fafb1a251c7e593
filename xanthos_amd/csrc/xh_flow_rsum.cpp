// Host-side planner of the REASSOCIATED ("tolerance") form of the dataflow routing kernel (k_mrtm_rsum, xh_mrtm_rsum.hip).
//
// Reference: mrtm.py:50-51 forms row i of UM.dot(F) as ((0 + F_a) + F_b) ... - F_i + F_x ... in stored (ascending column)
// order.  The bit-exact planner (xh_flow_plan.cpp) keeps that order, which is why its units gather up to six values per
// sub-step.  Here only the VALUE of the sum is kept (to rounding): every upstream neighbour of a cell -- either side of the
// diagonal -- passes a running sum along a chain of lanes, and a cell reads
//     A  the running sum of the LAST member of its children's chain   = sum of the flows of all its upstream neighbours
//     R  the running sum of the member in front of it in its own chain (its previous sibling)
// and stores R + (its own flow).  Two LDS reads per sub-step for every unit, whatever its rows look like; no plain form,
// no learning, no guard.  Same pieces / units / one-way streams / time skew as the bit-exact form: a lane that reads a
// value runs one level (two sub-steps) behind the lane -- or imported stream -- that produced it.
//
// What differs in the partition:
//   * children that are cut off as pieces of their own are chained ACROSS pieces: the outlet of sibling piece k takes the
//     stream of sibling piece k - 1 as its R and exports the running sum, so a cell imports at most ONE stream (the total
//     of all its cut-off children), which opens the chain of the children that stayed (or is the cell's A if none did);
//   * no row-shape rules in the cut or the packing: every unit has the same shape, so pieces are cut for full lanes only;
//   * the order of a chain is free: children with the tallest (transformed) subtrees go last, which keeps lane lags low.
//
// SINGLE-SUM plans (FlowPlanOptions::capable given; kernel side: xh_mrtm_wave_unit.h, SGL).  The pair {sum F, sum F2} of
// mrtm.py:51 / :66 is only ever needed by a cell that may fire itself AND has an upstream neighbour that may: a neighbour
// that does not fire has F2 = F, and a cell that cannot fire has S1 >= S2 >= 0 and uses the adjusted sum only.  So
//   X  = the cells that MAY FIRE: those that can by construction (velocity dt / length >= 1: `capable`) and, below every
//        cell of D0 = {capable with a capable upstream neighbour}, a HALO of `halo` cells: the reference's own corner
//        S1 >= 0 > S2 (mrtm.py:54, :66-69) leaves NEGATIVE storage in a D0 cell, its outflow turns negative, and cells
//        downstream fire that cannot by construction -- as far as the negative flows reach before the tributaries outweigh
//        them (the halo restarts at every capable cell it meets);
//   D  = the cells of X with an upstream neighbour in X;   P = D and the upstream neighbours in X of D's cells.
// The cells of P sit in PAIR units (pieces of their own, units of their own: unit_p bit 4), everything else in SINGLE units
// (bit 3 alone): their lanes pass ONE running sum, that of the adjusted flows, in 8-byte entries, and export {y, y}.  A
// pair unit that imports from single units therefore reads F = F2 -- exact, because those producers are not in X (an
// upstream neighbour in X of a D cell is in P) and a cell outside X does not fire as long as every flow it receives is
// >= 0, which holds by induction from non-negative runoff and initial storage (guarded in the kernel) once the negative
// flows of the D cells are confined: the EXIT lanes (lane_flags bit 1: cells of P whose downstream cell is not in P) are
// guarded, outflow >= 0 in every sub-step, and a trip routes the call again on the plan of pairs.  Chains of sibling
// pieces run single pieces first, pair pieces last (a single piece behind a pair piece would drop sum F).
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <string>
#include <thread>

#include "xh_flow_plan.h"

using xh_flow::Tree;
using xh_flow::tree_analyse;

namespace {

constexpr int LANES = 64;
constexpr int G_MAX = 16;
constexpr int W_MAX = 9;
constexpr int SK_P = 4;
constexpr unsigned SK_ZERO = 2u * LANES * 16u;
constexpr int H_MAX = 88;         // tallest piece (levels): lane lags stay below the shortest month's sub-steps (wave_launch)

struct RPart {
    std::vector<int> queue, piece, roots, psize, pimp, pdepth;
    std::vector<int> e_prod, e_cons, e_kind;      // stream: producer outlet, consumer cell, 0 = opens the chain of e_cons's children
                                                  // (or is its A), 1 = R of the outlet e_cons of a sibling piece
    std::vector<int> unit_of_piece, unit_cells, unit_imp, unit_out, unit_depth;
    std::vector<char> unit_cheap, unit_special, special, mayfire;      // special: the cell is in P; mayfire: in X (single-sum plans)
    int nunit = 0, nedge = 0, maxdepth = 0, n_cheap = 0;
    int n_special = -1;               // -1: plan of pairs (no `capable`); >= 0: single-sum plan with that many cells in P
};

void rsum_partition(const Tree &t, const FlowPlanOptions &opt, int cap, RPart &P, std::vector<int> &fold_leaf) {
    const int n = t.n;
    const std::vector<int> &ds = t.ds, &child = t.child, &child_ptr = t.child_ptr;
    P = RPart();
    std::vector<int> &queue = P.queue;
    queue.reserve(n);
    std::vector<int> left(t.nchild);
    for (int c = 0; c < n; ++c)
        if (t.ok[c] && t.nchild[c] == 0) queue.push_back(c);
    std::vector<int> dsu(n);
    std::iota(dsu.begin(), dsu.end(), 0);
    auto find = [&](int x) {
        while (dsu[x] != x) {
            dsu[x] = dsu[dsu[x]];
            x = dsu[x];
        }
        return x;
    };
    // single-sum plan: X (may fire), D, P (header)
    std::vector<char> &special = P.special, &mayfire = P.mayfire;
    special.assign(n, 0);
    mayfire.assign(n, 0);
    if (opt.capable) {
        P.n_special = 0;
        for (int v = 0; v < n; ++v) mayfire[v] = (t.ok[v] && opt.capable[v]) ? 1 : 0;
        const int H = std::max(opt.halo, 0);
        for (int v = 0; v < n; ++v) {
            if (!t.ok[v] || !opt.capable[v]) continue;
            bool d0 = false;
            for (int k = child_ptr[v]; k < child_ptr[v + 1] && !d0; ++k) d0 = opt.capable[child[k]] != 0;
            if (!d0) continue;
            int h = 0;
            for (int x = ds[v]; x >= 0; x = ds[x]) {
                if (opt.capable[x]) {
                    h = 0;                         // may leave negative storage itself: the halo starts again
                } else {
                    if (h == H) break;
                    ++h;
                }
                mayfire[x] = 1;
            }
        }
        for (int v = 0; v < n; ++v) {
            if (!mayfire[v]) continue;
            bool d = false;
            for (int k = child_ptr[v]; k < child_ptr[v + 1]; ++k) d = d || mayfire[child[k]];
            if (!d) continue;
            special[v] = 1;
            for (int k = child_ptr[v]; k < child_ptr[v + 1]; ++k)
                if (mayfire[child[k]]) special[child[k]] = 1;
        }
        for (int v = 0; v < n; ++v) P.n_special += special[v];
    }
    std::vector<int> open_cnt(n, 0), open_imp(n, 0), open_th(n, 0);
    std::vector<int> kids, kept, ths;
    std::vector<int> &roots = P.roots;
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        const int v = queue[qi];
        kids.assign(child.begin() + child_ptr[v], child.begin() + child_ptr[v + 1]);
        std::sort(kids.begin(), kids.end(), [&](int x, int y) { return open_cnt[x] != open_cnt[y] ? open_cnt[x] < open_cnt[y] : x < y; });
        // transformed height of v's open piece for a set of children kept: the chain runs tallest subtree last, member i of p
        // sits p - i levels above the last one; a stream (the total of the children cut off) opens the chain, one level more
        auto height = [&](const std::vector<int> &kp, bool any_cut) {
            ths.clear();
            for (int c : kp) ths.push_back(open_th[c]);
            std::sort(ths.begin(), ths.end());
            const int p = (int)ths.size();
            int h = any_cut ? (p > 0 ? p + 1 : 1) : 0;
            for (int i = 0; i < p; ++i) h = std::max(h, 1 + (p - 1 - i) + ths[i]);
            return h;
        };
        int total = 1, imp = 0;
        kept.clear();
        // imports of a piece: those of the children kept + one for all the children cut off; two are held back (that one, and
        // the sibling stream this piece takes on if it is cut off itself as a later member of a chain of pieces)
        for (int c : kids)
            if (special[v] == special[c] && total + open_cnt[c] <= cap && imp + open_imp[c] <= G_MAX - 2) {
                kept.push_back(c);
                total += open_cnt[c];
                imp += open_imp[c];
            }
        while (!kept.empty() && height(kept, kept.size() < kids.size()) > H_MAX) {      // (rare: a comb-shaped piece)
            size_t worst = 0;
            for (size_t i = 1; i < kept.size(); ++i)
                if (open_th[kept[i]] > open_th[kept[worst]]) worst = i;
            total -= open_cnt[kept[worst]];
            imp -= open_imp[kept[worst]];
            kept.erase(kept.begin() + (long)worst);
        }
        const bool any_cut = kept.size() < kids.size();
        for (int c : kids) {
            if (std::find(kept.begin(), kept.end(), c) != kept.end()) dsu[find(c)] = v;
            else roots.push_back(c);                   // c's piece is final; its outlet streams towards v
        }
        open_cnt[v] = total;
        open_imp[v] = imp + (any_cut ? 1 : 0);
        open_th[v] = height(kept, any_cut);
        if (ds[v] < 0) roots.push_back(v);
        else if (--left[ds[v]] == 0) queue.push_back(ds[v]);
    }

    // ---- pieces, stream edges (siblings chained), pipeline depth
    const int npiece = (int)roots.size();
    std::vector<int> piece_of_root(n, -1);
    for (int p = 0; p < npiece; ++p) piece_of_root[roots[p]] = p;
    P.piece.assign(n, -1);
    P.psize.assign(npiece, 0);
    P.pimp.assign(npiece, 0);
    P.pdepth.assign(npiece, 0);
    for (int v : queue) {
        const int q = piece_of_root[find(v)];
        P.piece[v] = q;
        P.psize[q]++;
    }
    std::vector<int> grp;
    for (int p = 0; p < npiece;) {                     // closing order: every producer of a piece comes before it
        const int v = ds[roots[p]];
        if (v < 0) {
            ++p;
            continue;
        }
        grp.clear();
        while (p < npiece && ds[roots[p]] == v) grp.push_back(p++);
        // the chain of sibling pieces: shallowest first (each member sits one pipeline level below the one before it)
        // (single-sum plans: single pieces first, pair pieces last -- a single piece behind a pair piece would drop sum F)
        auto rank = [&](int x) { return special[roots[x]] ? 1 : 0; };
        std::stable_sort(grp.begin(), grp.end(), [&](int x, int y) {
            if (rank(x) != rank(y)) return rank(x) < rank(y);
            return P.pdepth[x] != P.pdepth[y] ? P.pdepth[x] < P.pdepth[y] : P.psize[x] > P.psize[y];
        });
        for (size_t i = 0; i < grp.size(); ++i) {
            const bool last = i + 1 == grp.size();
            const int cons_piece = last ? P.piece[v] : grp[i + 1];
            P.e_prod.push_back(roots[grp[i]]);
            P.e_cons.push_back(last ? v : roots[grp[i + 1]]);
            P.e_kind.push_back(last ? 0 : 1);
            P.pimp[cons_piece]++;
            P.pdepth[cons_piece] = std::max(P.pdepth[cons_piece], P.pdepth[grp[i]] + 1);
        }
    }
    P.nedge = (int)P.e_prod.size();
    P.maxdepth = npiece ? *std::max_element(P.pdepth.begin(), P.pdepth.end()) : 0;

    // ---- folded leaves (opt.foldable).  A leaf that cannot fire is a one-fma recurrence, which the lane of its downstream cell
    //      can carry in registers (xh_mrtm_wave_unit.h, FOLD): no lane, no LDS slot, no level of lag -- at ~4 more fp64
    //      operations per sub-step for the whole unit of that lane.  Only units that nobody waits for and that wait for nobody
    //      can afford that for free, so only pieces WITHOUT STREAMS (whole small networks) fold: every cell of such a piece takes
    //      ONE of its foldable leaf children into its lane.  The pieces shrink; what this buys is units: 67,420 cells are 2.9 %
    //      more than the chip's lanes, and the ~40 SIMDs that had to hold two units each ended the kernel.
    fold_leaf.assign(n, -1);
    // fold_pieces(ok): every cell of a piece `ok` accepts takes ONE of its foldable leaf children (same piece) into its lane;
    // returns the lanes saved
    auto fold_pieces = [&](auto &&accept) {
        int saved = 0;
        for (int c : queue) {
            const int q = P.piece[c];
            if (q < 0 || fold_leaf[c] >= 0 || special[c] || !accept(q)) continue;      // (pair pieces carry no folded leaves)
            for (int k = child_ptr[c]; k < child_ptr[c + 1]; ++k) {
                const int l = child[k];
                if (opt.foldable[l] && t.nchild[l] == 0 && P.piece[l] == q) {
                    fold_leaf[c] = l;
                    P.psize[q]--;
                    P.piece[l] = -1;
                    ++saved;
                    break;
                }
            }
        }
        queue.erase(std::remove_if(queue.begin(), queue.end(), [&](int c) { return P.piece[c] < 0; }), queue.end());
        return saved;
    };
    if (opt.foldable) fold_pieces([&](int p) { return P.pimp[p] == 0 && ds[roots[p]] < 0; });
    // pieces that carry folded leaves may only sit in units WITHOUT IMPORTS (the kernel's FOLD variant), and must not fill the
    // free lanes of a unit that others wait for: the whole unit pays for the folded lanes
    std::vector<char> has_fold(npiece, 0);
    auto mark_folds = [&]() {
        std::fill(has_fold.begin(), has_fold.end(), 0);
        for (int c = 0; c < n; ++c)
            if (fold_leaf[c] >= 0) has_fold[P.piece[c]] = 1;
    };
    mark_folds();

    // ---- packing.  Pieces with a stream in or out: equal pipeline depth per unit (a unit then only ever waits for units
    //      strictly upstream or downstream of it), first-fit decreasing.  Whole small networks wait for nobody and fill free
    //      lanes anywhere; the `cheap` cheapest of them (single cells first: they read nothing; then the smallest) are kept
    //      together as units for the SIMDs that must hold two waves.
    auto has_out = [&](int p) { return ds[roots[p]] >= 0; };
    // (pair units pace a single-sum launch; with at most `pair_imports` streams they run the one-round variant of the kernel)
    const int pair_imp = std::min(std::max(opt.pair_imports, 1), G_MAX);
    int cheap = 0;
    // pack(): the partition for the pieces as they are now; returns the units beyond the SIMDs left for them
    auto pack = [&](bool cheap_units) {
    std::vector<int> dep, fre;
    for (int p = 0; p < npiece; ++p) (P.pimp[p] > 0 || has_out(p) ? dep : fre).push_back(p);
    std::stable_sort(dep.begin(), dep.end(), [&](int x, int y) {      // (pieces with folded leaves first: they share units)
        if (P.pdepth[x] != P.pdepth[y]) return P.pdepth[x] < P.pdepth[y];
        if (has_fold[x] != has_fold[y]) return has_fold[x] > has_fold[y];
        return P.psize[x] > P.psize[y];
    });
    std::vector<int> by_cost(fre);
    std::stable_sort(by_cost.begin(), by_cost.end(), [&](int x, int y) { return P.psize[x] < P.psize[y]; });
    int need = 0;
    for (int round = 0; round < 4; ++round) {
        P.unit_of_piece.assign(npiece, -1);
        P.unit_cells.clear();
        P.unit_imp.clear();
        P.unit_out.clear();
        P.unit_depth.clear();
        P.unit_cheap.clear();
        P.unit_special.clear();
        auto new_unit = [&](int depth, bool is_cheap) {
            P.unit_special.push_back(0);
            P.unit_cells.push_back(0);
            P.unit_imp.push_back(0);
            P.unit_out.push_back(0);
            P.unit_depth.push_back(depth);
            P.unit_cheap.push_back(is_cheap ? 1 : 0);
            return (int)P.unit_cells.size() - 1;
        };
        auto put_piece = [&](int p, int u) {
            P.unit_of_piece[p] = u;
            P.unit_cells[u] += P.psize[p];
            P.unit_imp[u] += P.pimp[p];
            P.unit_out[u] += has_out(p) ? 1 : 0;
        };
        {
            size_t first_open = 0;
            int cur_depth = -1;
            for (int p : dep) {
                if (P.pdepth[p] != cur_depth) {
                    cur_depth = P.pdepth[p];
                    first_open = P.unit_cells.size();
                }
                int u = -1;
                const char sp = special[roots[p]];      // pair pieces (single-sum plans) only share units with each other
                for (size_t b = first_open; b < P.unit_cells.size(); ++b)
                    if (P.unit_special[b] == sp && P.unit_cells[b] + P.psize[p] <= LANES && P.unit_imp[b] + P.pimp[p] <= (sp ? pair_imp : G_MAX) &&
                        P.unit_out[b] + (has_out(p) ? 1 : 0) <= G_MAX) {
                        u = (int)b;
                        break;
                    }
                if (u < 0) {
                    u = new_unit(cur_depth, false);
                    P.unit_special[u] = sp;
                }
                put_piece(p, u);
                while (first_open < P.unit_cells.size() && P.unit_cells[first_open] >= LANES) ++first_open;
            }
        }
        std::vector<char> taken(npiece, 0);
        {
            int made = 0, u = -1;
            for (int p : by_cost) {
                if (special[roots[p]] || has_fold[p]) continue;      // (cheap units: neither pair form nor folded leaves)
                if (u < 0 || P.unit_cells[u] + P.psize[p] > LANES) {
                    if (made == cheap) break;
                    u = new_unit(0, true);
                    ++made;
                }
                put_piece(p, u);
                taken[p] = 1;
            }
        }
        // the other free pieces: largest first, each into the fullest unit that still takes it
        std::vector<int> by_size;
        for (int p : fre)
            if (!taken[p]) by_size.push_back(p);
        std::stable_sort(by_size.begin(), by_size.end(), [&](int x, int y) { return P.psize[x] > P.psize[y]; });
        {
            std::vector<std::vector<int>> bucket(LANES + 1);
            for (int u = 0; u < (int)P.unit_cells.size(); ++u)
                if (!P.unit_cheap[u] && !P.unit_special[u]) bucket[LANES - P.unit_cells[u]].push_back(u);
            for (int p : by_size) {
                const int sz = P.psize[p];
                int u = -1;
                if (special[roots[p]]) {      // a whole small network in pair form: any pair unit with room, or one of its own
                    for (int b = 0; b < (int)P.unit_cells.size() && u < 0; ++b)
                        if (P.unit_special[b] && P.unit_cells[b] + sz <= LANES) u = b;
                    if (u < 0) {
                        u = new_unit(0, false);
                        P.unit_special[u] = 1;
                    }
                    put_piece(p, u);
                    continue;
                }
                for (int f = sz; f <= LANES && u < 0; ++f)
                    for (size_t i = bucket[f].size(); i-- > 0;) {
                        const int b = bucket[f][i];
                        if (has_fold[p] && (P.unit_imp[b] > 0 || P.unit_out[b] > 0)) continue;
                        u = b;
                        bucket[f].erase(bucket[f].begin() + (long)i);
                        break;
                    }
                if (u < 0) u = new_unit(0, false);
                put_piece(p, u);
                bucket[LANES - P.unit_cells[u]].push_back(u);
            }
        }
        P.nunit = (int)P.unit_cells.size();
        P.n_cheap = cheap;
        // (single-sum plans: every pair unit gets a CU of its own -- wave_claim --, whose other three SIMDs stay empty)
        int npair = 0;
        for (char c : P.unit_special) npair += c ? 1 : 0;
        const int simds_left = opt.simds - (opt.simds >= 64 ? std::min(3 * npair, opt.simds / 4) : 0);
        need = opt.simds > 0 ? std::max(P.nunit - simds_left, 0) : 0;
        if (need <= cheap || !cheap_units) break;
        cheap = need + (round > 0 ? 2 : 0);
    }
    return need;
    };
    // More units than SIMDs: lanes can be saved where folding costs least -- in pieces WITHOUT IMPORTS (the upstream ends of the
    // rivers: many leaves, and their units, the cheapest of the launch, stay cheaper than the pair units that pace it even
    // with the folded leaves' four operations), the pieces with the most leaves to fold first, until the units fit.  What is
    // still missing then is covered the old way: cheap units of single cells for the SIMDs that must hold two.
    int need = pack(false);
    for (int pass = 0; pass < 4 && need > 0 && opt.foldable; ++pass) {      // (units only merge within a pipeline level: a pass may fall short)
        std::vector<int> gain(npiece, 0);
        for (int c : queue) {
            const int q = P.piece[c];
            if (q < 0 || special[c] || fold_leaf[c] >= 0 || P.pimp[q] != 0) continue;
            for (int k = child_ptr[c]; k < child_ptr[c + 1]; ++k)
                if (opt.foldable[child[k]] && t.nchild[child[k]] == 0 && P.piece[child[k]] == q) {
                    gain[q]++;
                    break;
                }
        }
        std::vector<int> order;
        for (int p = 0; p < npiece; ++p)
            if (gain[p] > 0) order.push_back(p);
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return gain[x] > gain[y]; });
        // (a unit's worth of lanes more than the count says: the packing is not perfect)
        const int want = (need + 1) * LANES + LANES / 2;
        std::vector<char> chosen(npiece, 0);
        int got = 0;
        for (int p : order) {
            if (got >= want) break;
            chosen[p] = 1;
            got += gain[p];
        }
        if (got == 0) break;
        fold_pieces([&](int p) { return chosen[p] != 0; });
        mark_folds();
        need = pack(false);
    }
    if (need > 0) pack(true);
}

}  // namespace

int flow_rsum_planner_version() { return 8; }

int flow_tables_build_rsum(int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign, const int *comp,
                           int ncomp, const FlowPlanOptions &opt, std::vector<char> &handled, FlowTables &out,
                           std::string &err) {
    out = FlowTables();
    out.rsum = true;
    handled.assign(n, 0);
    if (n == 0) return 0;
    Tree t;
    tree_analyse(n, indptr, indices, sign, comp, ncomp, t);
    const std::vector<int> &ds = t.ds;

    RPart P;
    std::vector<int> fold_leaf;                        // cell -> the leaf its lane carries, or -1
    {
        static const int caps[] = {LANES, 56, 48, 44, 40, 36, 32, 28, 24};
        constexpr int NCAP = (int)(sizeof(caps) / sizeof(caps[0]));
        std::vector<RPart> cand(opt.piece_cap > 0 ? 1 : NCAP);
        std::vector<std::vector<int>> cand_fold(cand.size());
        if (cand.size() == 1 || n < 4096) {
            for (size_t k = 0; k < cand.size(); ++k)
                rsum_partition(t, opt, opt.piece_cap > 0 ? std::min(opt.piece_cap, LANES) : caps[k], cand[k], cand_fold[k]);
        } else {
            std::vector<std::thread> pool;
            for (size_t k = 0; k < cand.size(); ++k) pool.emplace_back([&, k] { rsum_partition(t, opt, caps[k], cand[k], cand_fold[k]); });
            for (auto &th : pool) th.join();
        }
        bool have = false;
        long best_score = 0;
        for (size_t k = 0; k < cand.size(); ++k) {
            RPart &Q = cand[k];
            int indep = 0;
            for (int u = 0; u < Q.nunit; ++u) indep += (Q.unit_imp[u] == 0 && Q.unit_out[u] == 0) ? 1 : 0;
            int npair = 0;
            for (char c : Q.unit_special) npair += c ? 1 : 0;
            const int extra = opt.simds > 0 ? std::max(Q.nunit - (opt.simds - (opt.simds >= 64 ? std::min(3 * npair, opt.simds / 4) : 0)), 0) : 0;
            // units beyond the SIMD count share a SIMD, and only units without streams may (the claim order below); then as
            // few units as possible that pay for folded leaves although others wait for them; then few units, few streams
            std::vector<char> ufold(Q.nunit, 0);
            for (int c = 0; c < n; ++c)
                if (cand_fold[k][c] >= 0) ufold[Q.unit_of_piece[Q.piece[c]]] = 1;
            int fold_streams = 0;
            for (int u = 0; u < Q.nunit; ++u) fold_streams += (ufold[u] && (Q.unit_imp[u] > 0 || Q.unit_out[u] > 0)) ? 1 : 0;
            const long score = 1000000L * std::max(2 * extra - indep, 0) + 20000L * fold_streams + 1000L * Q.nunit + Q.nedge / 8;
            if (opt.debug)
                fprintf(stderr, "flow plan (reassociated): piece capacity %d -> %d units, %d streams, %d units without streams, depth %d\n",
                        opt.piece_cap > 0 ? opt.piece_cap : caps[k], Q.nunit, Q.nedge, indep, Q.maxdepth + 1);
            if (!have || score < best_score) {
                std::swap(P, Q);
                std::swap(fold_leaf, cand_fold[k]);
                best_score = score;
                have = true;
            }
        }
    }
    const std::vector<int> &piece = P.piece, &roots = P.roots, &unit_of_piece = P.unit_of_piece;
    const int nunit = P.nunit, nedge = P.nedge;
    if (nunit == 0) return 0;
    const int64_t ts = (int64_t)nunit * LANES;

    // ---- slots and imported entries
    out.cell_of_slot.assign(ts, -1);
    out.export_edge.assign(ts, -1);
    out.ghost_edge.assign(ts, -1);
    out.ghost_prod.assign(ts, 0);
    out.edge_cons_unit.assign(nedge, 0);
    std::vector<int> slot_of_cell(n, -1), fill(nunit, 0), gfill(nunit, 0), edge_ghost(nedge, 0);
    for (int c = 0; c < n; ++c)
        if (piece[c] >= 0) {
            const int u = unit_of_piece[piece[c]];
            const int s = fill[u]++;
            if (s >= LANES) {
                err = "flow plan (reassociated): a unit holds more than 64 cells";
                return -1;
            }
            out.cell_of_slot[(int64_t)u * LANES + s] = c;
            slot_of_cell[c] = s;
            handled[c] = 1;
        }
    for (int c = 0; c < n; ++c)
        if (fold_leaf[c] >= 0) {
            if (piece[c] < 0) {
                err = "flow plan (reassociated): a folded leaf lost its carrier";
                return -1;
            }
            if (out.fold_of_slot.empty()) out.fold_of_slot.assign(ts, -1);
            out.fold_of_slot[(int64_t)unit_of_piece[piece[c]] * LANES + slot_of_cell[c]] = fold_leaf[c];
            handled[fold_leaf[c]] = 1;
            out.n_folded++;
        }
    std::vector<int> edge_in0(n, -1), edge_in1(n, -1);      // per cell: the stream that feeds its children's sum / its own R
    for (int ed = 0; ed < nedge; ++ed) {
        const int cc = P.e_cons[ed], u = unit_of_piece[piece[cc]];
        const int g = gfill[u]++;
        if (g >= G_MAX) {
            err = "flow plan (reassociated): a unit imports more than 16 streams";
            return -1;
        }
        out.edge_cons_unit[ed] = u;
        edge_ghost[ed] = g;
        out.ghost_edge[(int64_t)u * LANES + g] = ed;
        out.ghost_prod[(int64_t)u * LANES + g] = P.e_prod[ed];
        const int pc = P.e_prod[ed];
        out.export_edge[(int64_t)unit_of_piece[piece[pc]] * LANES + slot_of_cell[pc]] = ed;
        (P.e_kind[ed] == 0 ? edge_in0 : edge_in1)[cc] = ed;
    }

    // ---- chains, heights.  Bottom-up: transformed height of every cell's subtree inside its piece; top-down: levels.
    std::vector<int> th(n, 0), hgt(n, 0), a_src(n, -1), r_src(n, -1);      // sources: cell id, or -2 - edge, or -1 = zero
    std::vector<int> kids;
    for (int c : P.queue) {
        kids.clear();
        for (int k = t.child_ptr[c]; k < t.child_ptr[c + 1]; ++k)
            if (piece[t.child[k]] >= 0 && piece[t.child[k]] == piece[c]) kids.push_back(t.child[k]);      // (a folded leaf has no piece)
        std::stable_sort(kids.begin(), kids.end(), [&](int x, int y) { return th[x] < th[y]; });
        const int p = (int)kids.size();
        int h = edge_in0[c] >= 0 ? (p > 0 ? p + 1 : 1) : 0;
        for (int i = 0; i < p; ++i) h = std::max(h, 1 + (p - 1 - i) + th[kids[i]]);
        th[c] = h;
        // (levels are handed out below, once the parent's level is known; the chain order is fixed here)
        for (int i = 0; i < p; ++i) r_src[kids[i]] = i > 0 ? kids[i - 1] : (edge_in0[c] >= 0 ? -2 - edge_in0[c] : -1);
        a_src[c] = p > 0 ? kids[p - 1] : (edge_in0[c] >= 0 ? -2 - edge_in0[c] : -1);
        // hgt[] temporarily holds a member's offset above the last member of its chain
        for (int i = 0; i < p; ++i) hgt[kids[i]] = p - 1 - i;
    }
    std::vector<int> unit_h(nunit, 0), edge_reader_h(nedge, 0);
    for (size_t qi = P.queue.size(); qi-- > 0;) {       // downstream cells first
        const int c = P.queue[qi];
        if (piece[c] < 0) continue;
        const bool root = roots[piece[c]] == c;
        if (root) {
            hgt[c] = 0;
            if (edge_in1[c] >= 0) r_src[c] = -2 - edge_in1[c];
        } else {
            hgt[c] = hgt[ds[c]] + 1 + hgt[c];
        }
        int &uh = unit_h[unit_of_piece[piece[c]]];
        uh = std::max(uh, hgt[c]);
    }
    for (int c = 0; c < n; ++c) {                       // an imported value sits one level above the lane that reads it
        if (piece[c] < 0) continue;
        for (int src : {a_src[c], r_src[c]})
            if (src <= -2) {
                const int ed = -2 - src;
                edge_reader_h[ed] = hgt[c];
                int &uh = unit_h[out.edge_cons_unit[ed]];
                uh = std::max(uh, hgt[c] + 1);
            }
    }

    out.lag.assign(ts, 0);
    out.ghost_lag.assign(ts, 0);
    out.unit_p.assign(nunit, 0x400);
    out.unit_lmax.assign(nunit, 0);
    out.unit_glmax.assign(nunit, 0);
    out.ent2.assign((size_t)2 * SK_P * ts, SK_ZERO);
    out.eprev.assign(ts, SK_ZERO);
    out.ent.assign((size_t)W_MAX * ts, SK_ZERO);        // (the lock-step kernel's table: this plan is never routed by it)
    out.unit_terms.assign(nunit, 1);
    for (int u = 0; u < nunit; ++u) out.unit_lmax[u] = (2 * unit_h[u] + 15) & ~15;
    auto entry_of = [&](int src, int u) -> unsigned {
        if (src == -1) return SK_ZERO;
        if (src <= -2) return (unsigned)(LANES + edge_ghost[-2 - src]) * 16u;
        (void)u;
        return (unsigned)slot_of_cell[src] * 16u;
    };
    for (int c = 0; c < n; ++c) {
        if (piece[c] < 0) continue;
        const int u = unit_of_piece[piece[c]];
        const int64_t slot = (int64_t)u * LANES + slot_of_cell[c];
        out.lag[slot] = out.unit_lmax[u] - 2 * hgt[c];
        out.ent2[(size_t)slot] = entry_of(a_src[c], u);
        out.eprev[slot] = entry_of(r_src[c], u);
        if (a_src[c] != -1) out.unit_p[u] |= 1;
        if (r_src[c] != -1) out.unit_p[u] |= 2;
        if (fold_leaf[c] >= 0) out.unit_p[u] |= 4;                  // the unit carries folded leaves: the FOLD variant of the kernel
    }
    out.n_special = P.n_special;
    out.n_pair_units = 0;
    if (P.n_special >= 0)                                           // single-sum plan: bit 3; bit 4: a pair unit (the cells of P)
        for (int u = 0; u < nunit; ++u) {
            out.unit_p[u] |= P.unit_special[u] ? 24 : 8;
            out.n_pair_units += P.unit_special[u] ? 1 : 0;
        }
    for (int ed = 0; ed < nedge; ++ed) {
        const int u = out.edge_cons_unit[ed];
        const int gl = out.unit_lmax[u] - 2 * (edge_reader_h[ed] + 1);
        out.ghost_lag[(int64_t)u * LANES + edge_ghost[ed]] = gl;
        out.unit_glmax[u] = std::max(out.unit_glmax[u], gl);
    }
    out.lane_flags.assign(ts, 0);
    if (P.n_special >= 0)      // bit 0: the cell may fire (X); bit 1: an exit lane (a cell of P whose downstream cell is not in P)
        for (int64_t sl = 0; sl < ts; ++sl) {
            const int c = out.cell_of_slot[sl];
            if (c < 0) continue;
            out.lane_flags[sl] = (unsigned char)((P.mayfire[c] ? 1 : 0) | ((P.special[c] && ds[c] >= 0 && !P.special[ds[c]]) ? 2 : 0));
        }

    std::vector<int> unit_exp(nunit, 0);
    for (int ed = 0; ed < nedge; ++ed) unit_exp[unit_of_piece[piece[P.e_prod[ed]]]]++;
    {   // ---- the claim list (top of the kernel): units without streams by rising cost, then the others.  The SIMDs that hold
        //      two waves get the cheapest units without streams and, with issue priority, the next ones.
        std::vector<int> cost(nunit);
        for (int u = 0; u < nunit; ++u)
            cost[u] = 100 + 25 * ((out.unit_p[u] & 1) + ((out.unit_p[u] >> 1) & 1)) + (P.unit_imp[u] > 0 ? 15 : 0) + (unit_exp[u] > 0 ? 15 : 0);
        auto coupled = [&](int u) { return P.unit_imp[u] > 0 || unit_exp[u] > 0; };
        out.unit_order.resize(nunit);
        std::iota(out.unit_order.begin(), out.unit_order.end(), 0);
        // (single-sum plans: the pair units close the list -- the kernel gives the first of them a CU of their own)
        auto pairu = [&](int u) { return P.n_special >= 0 && P.unit_special[u]; };
        std::stable_sort(out.unit_order.begin(), out.unit_order.end(), [&](int x, int y) {
            if (pairu(x) != pairu(y)) return !pairu(x);
            if (coupled(x) != coupled(y)) return !coupled(x);
            if (cost[x] != cost[y]) return cost[x] < cost[y];
            return P.unit_cells[x] < P.unit_cells[y];
        });
    }

    out.skew_ok = true;
    out.skew_lmax = *std::max_element(out.unit_lmax.begin(), out.unit_lmax.end());
    {
        int span = 1;
        for (int ed = 0; ed < nedge; ++ed) span = std::max(span, P.pdepth[piece[P.e_cons[ed]]] - P.pdepth[piece[P.e_prod[ed]]]);
        out.skew_span = span;
    }
    out.n_units = nunit;
    out.n_edges = nedge;
    out.depth = P.maxdepth + 1;
    out.n_cells = (int)std::count(handled.begin(), handled.end(), (char)1);
    out.max_imports = *std::max_element(P.unit_imp.begin(), P.unit_imp.end());
    out.max_exports = *std::max_element(unit_exp.begin(), unit_exp.end());
    out.edge_prod_cell = P.e_prod;
    out.edge_cons_cell = P.e_cons;
    out.unit_depth = P.unit_depth;
    out.piece_of_cell = piece;
    out.unit_of_cell.assign(n, -1);
    for (int c = 0; c < n; ++c)
        if (piece[c] >= 0) out.unit_of_cell[c] = unit_of_piece[piece[c]];
    out.height_of_cell = hgt;
    out.ds = ds;

    if (opt.debug) {
        std::vector<int> hs(4, 0), hl(14, 0), hi(6, 0);
        int indep = 0;
        for (int u = 0; u < nunit; ++u) {
            hs[out.unit_p[u] & 3]++;
            hl[std::min(out.unit_lmax[u] / 16, 13)]++;
            hi[P.unit_imp[u] == 0 ? 0 : P.unit_imp[u] <= 2 ? 1 : P.unit_imp[u] <= 4 ? 2 : P.unit_imp[u] <= 8 ? 3 : 4]++;
            indep += (P.unit_imp[u] == 0 && unit_exp[u] == 0) ? 1 : 0;
        }
        fprintf(stderr, "flow plan (reassociated): %d units (%d without streams, %d kept cheap), %d pieces, %d streams, depth %d, %lld of %lld lanes used\n",
                nunit, indep, P.n_cheap, (int)roots.size(), nedge, P.maxdepth + 1, (long long)out.n_cells, (long long)ts);
        {
            int nfu = 0, nfi = 0;
            for (int u = 0; u < nunit; ++u) {
                nfu += (out.unit_p[u] & 4) ? 1 : 0;
                nfi += ((out.unit_p[u] & 4) && (P.unit_imp[u] > 0 || unit_exp[u] > 0)) ? 1 : 0;
            }
            fprintf(stderr, "  folded leaves: %d, in %d units (of them with streams: %d)\n", out.n_folded, nfu, nfi);
        }
        {
            int nsu = 0;
            for (int u = 0; u < nunit; ++u) nsu += (out.unit_p[u] & 16) ? 1 : 0;
            int nx = 0;
            for (int c = 0; c < n; ++c) nx += (P.n_special >= 0 && P.mayfire[c]) ? 1 : 0;
            int quiet = 0;      // single units without a cell that may fire
            for (int u = 0; u < nunit && P.n_special >= 0; ++u) {
                bool any = (out.unit_p[u] & 16) != 0;
                for (int k = 0; k < LANES && !any; ++k) any = (out.lane_flags[(int64_t)u * LANES + k] & 1) != 0;
                quiet += any ? 0 : 1;
            }
            for (int u = 0; u < nunit && P.n_special >= 0; ++u)
                if (out.unit_p[u] & 16)
                    fprintf(stderr, "  pair unit %d: %d cells, %d imports, %d outlets, depth %d, lmax %d\n", u, P.unit_cells[u], P.unit_imp[u], unit_exp[u],
                            P.unit_depth[u], out.unit_lmax[u]);
            fprintf(stderr, "  single-sum plan: %s, %d cells may fire, %d of them in pair form in %d units; %d single units hold no cell that may fire\n",
                    P.n_special >= 0 ? "yes" : "no", nx, std::max(P.n_special, 0), nsu, quiet);
        }
        fprintf(stderr, "  units by reads (none, A, R, A+R): %d %d %d %d\n  units by lmax/16 (0..13+):", hs[0], hs[1], hs[2], hs[3]);
        for (int k = 0; k < 14; ++k) fprintf(stderr, " %d", hl[k]);
        fprintf(stderr, "\n  units by imports (0, <=2, <=4, <=8, more): %d %d %d %d %d\n  longest stream jump: %d levels\n", hi[0], hi[1],
                hi[2], hi[3], hi[4], out.skew_span);
    }
    return 0;
}

// Invariants of a reassociated plan, re-derived from the tables alone: every handled cell in exactly one slot; streams run
// strictly down the pipeline; <= 16 imports / outlets per unit; every value a lane reads was produced exactly one level
// (two sub-steps) earlier; the running sum a cell reads as its inflow expands -- through chains of lanes and of pieces -- to
// exactly the upstream neighbours of its CSR row (mrtm.py:50-51), each once; every lane's value is read by exactly one reader.
std::string flow_tables_check_rsum(int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign,
                                   const std::vector<char> &handled, const FlowTables &t) {
    if (!t.rsum) return "not a reassociated plan";
    if (t.n_units == 0) {
        for (int c = 0; c < n; ++c)
            if (handled[c]) return "handled cell without units";
        return "";
    }
    const int64_t ts = (int64_t)t.n_units * LANES;
    if ((int64_t)t.cell_of_slot.size() != ts || (int64_t)t.lag.size() != ts || (int64_t)t.eprev.size() != ts ||
        (int64_t)t.ent2.size() != 2 * SK_P * ts || (int)t.unit_p.size() != t.n_units || (int64_t)t.ghost_edge.size() != ts ||
        (int64_t)t.ghost_lag.size() != ts || (int64_t)t.export_edge.size() != ts || (int)t.unit_lmax.size() != t.n_units ||
        (int)t.unit_order.size() != t.n_units || (int)t.edge_cons_unit.size() != t.n_edges ||
        (int)t.edge_prod_cell.size() != t.n_edges || (int)t.edge_cons_cell.size() != t.n_edges || (int)t.ds.size() != n ||
        (int)t.unit_depth.size() != t.n_units)
        return "table sizes";
    std::vector<int> slot_of(n, -1);
    for (int64_t s = 0; s < ts; ++s) {
        const int c = t.cell_of_slot[s];
        if (c < 0) continue;
        if (c >= n || !handled[c]) return "slot holds a cell that is not handled";
        if (slot_of[c] >= 0) return "cell " + std::to_string(c) + " sits in two slots";
        slot_of[c] = (int)s;
    }
    // folded leaves: carried by the lane of their downstream cell, no slot of their own, no upstream neighbours
    std::vector<int> carrier(n, -1);
    if (!t.fold_of_slot.empty()) {
        if ((int64_t)t.fold_of_slot.size() != ts) return "fold table size";
        for (int64_t s = 0; s < ts; ++s) {
            const int l = t.fold_of_slot[s];
            if (l < 0) continue;
            const int c = t.cell_of_slot[s];
            if (c < 0 || l >= n || !handled[l] || slot_of[l] >= 0 || carrier[l] >= 0) return "folded leaf " + std::to_string(l);
            if (t.ds[l] != c) return "folded leaf " + std::to_string(l) + " is not carried by its downstream cell";
            if (indptr[l + 1] - indptr[l] != 1) return "folded cell " + std::to_string(l) + " is not a leaf";
            if (!(t.unit_p[s / LANES] & 4)) return "unit shape lacks the fold flag";
            carrier[l] = c;
        }
    }
    for (int c = 0; c < n; ++c)
        if (handled[c] && slot_of[c] < 0 && carrier[c] < 0) return "handled cell " + std::to_string(c) + " has no slot";
    {
        std::vector<char> seen(t.n_units, 0);
        for (int u : t.unit_order) {
            if (u < 0 || u >= t.n_units || seen[u]) return "claim list is not a permutation of the units";
            seen[u] = 1;
        }
    }
    std::vector<int> imp(t.n_units, 0), exp(t.n_units, 0);
    for (int ed = 0; ed < t.n_edges; ++ed) {
        const int pc = t.edge_prod_cell[ed], cc = t.edge_cons_cell[ed];
        if (pc < 0 || pc >= n || cc < 0 || cc >= n || slot_of[pc] < 0 || slot_of[cc] < 0) return "stream between cells that are not routed";
        // a stream follows a flow edge, or links the outlets of two sibling pieces
        if (t.ds[pc] != cc && !(t.ds[pc] >= 0 && t.ds[pc] == t.ds[cc])) return "stream follows neither a flow edge nor a sibling link";
        const int pu = slot_of[pc] / LANES, cu = slot_of[cc] / LANES;
        if (t.export_edge[slot_of[pc]] != ed) return "export_edge of the producer";
        if (t.edge_cons_unit[ed] != cu) return "edge_cons_unit";
        if (t.unit_depth[pu] >= t.unit_depth[cu]) return "stream from depth " + std::to_string(t.unit_depth[pu]) + " to depth " + std::to_string(t.unit_depth[cu]);
        imp[cu]++;
        exp[pu]++;
    }
    for (int u = 0; u < t.n_units; ++u) {
        if (imp[u] > G_MAX || exp[u] > G_MAX) return "more than 16 imports / outlets in unit " + std::to_string(u);
        int g = 0;
        for (int k = 0; k < LANES; ++k)
            if (t.ghost_edge[(int64_t)u * LANES + k] >= 0) {
                if (k != g) return "ghost entries are not packed from 0";
                if (t.ghost_edge[(int64_t)u * LANES + k] >= t.n_edges || t.edge_cons_unit[t.ghost_edge[(int64_t)u * LANES + k]] != u)
                    return "ghost entry of a stream that does not end in the unit";
                if (t.ghost_prod[(int64_t)u * LANES + k] != t.edge_prod_cell[t.ghost_edge[(int64_t)u * LANES + k]]) return "ghost_prod";
                ++g;
            }
        if (g != imp[u]) return "ghost count";
        if (t.unit_lmax[u] & 15) return "unit lag not a multiple of 16";
        if ((t.unit_p[u] & ~31) != 0x400) return "shape word of unit " + std::to_string(u);
        if (t.n_special >= 0 ? !(t.unit_p[u] & 8) : (t.unit_p[u] & 24) != 0) return "single-sum flags of unit " + std::to_string(u);
    }
    // what a value stands for: the cells whose flows it sums.  expand(entry) appends them.
    std::vector<int> readers_lane(ts, 0), readers_ghost(t.n_edges, 0);
    auto expand = [&](int u, unsigned off, std::vector<int> &terms, int depth, auto &&self) -> bool {
        if (off == SK_ZERO) return true;
        if (off % 16u) return false;
        const int e = (int)(off / 16u);
        if (e >= 2 * LANES || depth > 4 * LANES + t.n_edges) return false;
        if (e >= LANES) {                                   // imported: whatever the producing outlet exports
            const int ed = t.ghost_edge[(int64_t)u * LANES + (e - LANES)];
            if (ed < 0) return false;
            const int ps = slot_of[t.edge_prod_cell[ed]];
            return self(ps / LANES, (unsigned)(ps % LANES) * 16u, terms, depth + 1, self);
        }
        const int64_t s = (int64_t)u * LANES + e;
        const int c = t.cell_of_slot[s];
        if (c < 0) return false;
        if (!self(u, t.eprev[s], terms, depth + 1, self)) return false;
        terms.push_back(c);
        return true;
    };
    auto lag_of = [&](int u, unsigned off) {
        const int e = (int)(off / 16u);
        return e >= LANES ? t.ghost_lag[(int64_t)u * LANES + (e - LANES)] : t.lag[(int64_t)u * LANES + e];
    };
    auto note_reader = [&](int u, unsigned off) {
        if (off == SK_ZERO) return;
        const int e = (int)(off / 16u);
        if (e >= LANES) readers_ghost[t.ghost_edge[(int64_t)u * LANES + (e - LANES)]]++;
        else readers_lane[(int64_t)u * LANES + e]++;
    };
    std::vector<int> terms, want;
    for (int c = 0; c < n; ++c) {
        if (!handled[c] || slot_of[c] < 0) continue;
        const int s = slot_of[c], u = s / LANES;
        if (t.lag[s] < 0 || t.lag[s] > t.unit_lmax[u] || (t.lag[s] & 1)) return "lag of cell " + std::to_string(c);
        const unsigned a = t.ent2[(size_t)s], r = t.eprev[s];
        for (int w = 1; w < 2 * SK_P; ++w)
            if (t.ent2[(size_t)w * ts + s] != SK_ZERO) return "a reassociated plan reads one inflow entry per cell";
        for (unsigned off : {a, r}) {
            if (off == SK_ZERO) continue;
            if (off % 16u || off / 16u >= 2u * LANES) return "bad table entry at cell " + std::to_string(c);
            const int e = (int)(off / 16u);
            if (e >= LANES ? t.ghost_edge[(int64_t)u * LANES + (e - LANES)] < 0 : t.cell_of_slot[(int64_t)u * LANES + e] < 0)
                return "cell " + std::to_string(c) + " reads an empty entry";
            if (t.lag[s] - lag_of(u, off) != 2) return "a value read is not two iterations old at cell " + std::to_string(c);
            note_reader(u, off);
        }
        if (a != SK_ZERO && !(t.unit_p[u] & 1)) return "unit shape lacks the inflow read";
        if (r != SK_ZERO && !(t.unit_p[u] & 2)) return "unit shape lacks the chain read";
        terms.clear();
        if (!expand(u, a, terms, 0, expand)) return "bad chain at cell " + std::to_string(c);
        if (t.n_special >= 0) {
            // single-sum plan (lane_flags bit 0: the cell may fire): a cell that may fire with an upstream neighbour that may
            // sits in a pair unit, and so do those neighbours; along its chain no value of a single unit follows one of a pair unit
            if ((int64_t)t.lane_flags.size() != ts) return "lane flags";
            const bool pair_u = (t.unit_p[u] & 16) != 0;
            if (pair_u && !(t.lane_flags[s] & 1)) return "cell " + std::to_string(c) + " of a pair unit is not one that may fire";
            if (pair_u && !t.fold_of_slot.empty() && t.fold_of_slot[s] >= 0) return "folded leaf in a pair unit";
            int nx = 0;
            bool seen_pair = false, order_ok = true;
            for (int x : terms) {
                const bool xp = (t.unit_p[slot_of[x] / LANES] & 16) != 0;
                if ((t.lane_flags[slot_of[x]] & 1)) {
                    ++nx;
                    if ((t.lane_flags[s] & 1) && !xp) return "neighbour " + std::to_string(x) + " of cell " + std::to_string(c) + " may fire and is not in a pair unit";
                }
                if (seen_pair && !xp) order_ok = false;
                seen_pair = seen_pair || xp;
            }
            if ((t.lane_flags[s] & 1) && nx > 0) {
                if (!pair_u) return "cell " + std::to_string(c) + " may fire, has a neighbour that may, and sits in a single unit";
                if (!order_ok) return "chain of cell " + std::to_string(c) + ": a single unit's value behind a pair unit's";
            }
            const bool exit_want = pair_u && t.ds[c] >= 0 && slot_of[t.ds[c]] >= 0 && !(t.unit_p[slot_of[t.ds[c]] / LANES] & 16);
            if (((t.lane_flags[s] & 2) != 0) != exit_want) return "exit flag of cell " + std::to_string(c);
        }
        if (!t.fold_of_slot.empty() && t.fold_of_slot[s] >= 0) terms.push_back(t.fold_of_slot[s]);      // the leaf the lane carries itself
        want.clear();
        for (int64_t j = indptr[c]; j < indptr[c + 1]; ++j) {
            if (indices[j] == c) {
                if (sign[j] >= 0) return "diagonal sign";
            } else {
                want.push_back(indices[j]);
            }
        }
        std::sort(terms.begin(), terms.end());
        std::sort(want.begin(), want.end());
        if (terms != want) return "inflow of cell " + std::to_string(c) + " does not expand to its upstream neighbours";
    }
    // every value has exactly one reader (a lane's running sum: the next member or the fed cell; none for a cell without a
    // downstream cell), and every outlet with a stream is an exported lane
    for (int c = 0; c < n; ++c) {
        if (!handled[c] || slot_of[c] < 0) continue;
        const int s = slot_of[c];
        const int want_readers = t.ds[c] >= 0 ? 1 : 0;
        const int have = readers_lane[s] + (t.export_edge[s] >= 0 ? 1 : 0);
        if (have != want_readers) return "cell " + std::to_string(c) + " has " + std::to_string(have) + " readers";
    }
    for (int ed = 0; ed < t.n_edges; ++ed)
        if (readers_ghost[ed] != 1) return "stream " + std::to_string(ed) + " has " + std::to_string(readers_ghost[ed]) + " readers";
    return "";
}
