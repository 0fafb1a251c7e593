// Host-side planner of the dataflow routing kernels (see xh_flow_plan.h).  Three steps:
//   tree_analyse    which networks are plain trees; children lists; row shapes
//   make_partition  bottom-up cut into connected pieces, pieces packed into units of 64 lanes (one partition per
//                   piece capacity tried; the caller keeps the best)
//   emit_tables     slots, ghosts, gather offsets, lane lags, chains, claim order
//
// The dependency dS_i/dt = sum_{j upstream of i} F_j - F_i + lateral_i (mrtm.py:50-51) runs one way, so a tributary can
// be integrated ahead of the river it joins: pieces are linked by one-way streams, units of equal pipeline depth only
// ever wait on units strictly upstream (data) or downstream (ring space) of themselves.
#include "xh_flow_plan.h"

#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <numeric>
#include <thread>
#include <type_traits>
#include <utility>

using xh_flow::Tree;
using xh_flow::tree_analyse;

namespace {

constexpr int W_MAX = 9;          // terms per row: 8 D8 neighbours + the diagonal
constexpr int LANES = 64;         // cells per unit (one per lane)
constexpr int G_MAX = 16;         // imported streams, and outlets, per unit (two block-transfer rounds of 8)
constexpr int NPAIR = 2 * LANES + 1;
constexpr int SK_P = 4;           // row terms either side of the diagonal in the time-skewed layout
constexpr unsigned SK_ZERO = 2u * LANES * 16u;

}  // namespace

void xh_flow::tree_analyse(int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign, const int *comp, int ncomp,
                  Tree &t) {
    t.n = n;
    t.indptr = indptr;
    t.indices = indices;
    t.sign = sign;
    // ---- which networks are plain trees: rows are {-1 on the diagonal, +1 elsewhere}, every cell feeds <= 1 row,
    //      at most W_MAX terms per row, no cycle
    t.ds.assign(n, -1);
    std::vector<char> comp_ok(ncomp, 1);
    for (int r = 0; r < n; ++r) {
        int ndiag = 0;
        if (indptr[r + 1] - indptr[r] > W_MAX) comp_ok[comp[r]] = 0;
        for (int64_t j = indptr[r]; j < indptr[r + 1]; ++j) {
            const int c = indices[j];
            if (sign[j] < 0) {
                if (c == r) ++ndiag;
                else comp_ok[comp[r]] = 0;
            } else {
                if (c == r || t.ds[c] >= 0) comp_ok[comp[r]] = 0;
                t.ds[c] = r;
            }
        }
        if (ndiag != 1) comp_ok[comp[r]] = 0;
    }
    {   // cycles: follow the downstream pointers with three colours
        std::vector<char> colour(n, 0);
        std::vector<int> path;
        for (int s = 0; s < n; ++s) {
            if (colour[s]) continue;
            path.clear();
            int v = s;
            while (v >= 0 && colour[v] == 0) {
                colour[v] = 1;
                path.push_back(v);
                v = t.ds[v];
            }
            if (v >= 0 && colour[v] == 1) comp_ok[comp[v]] = 0;       // ran into the current path: a cycle
            for (int p : path) colour[p] = 2;
        }
    }
    t.ok.assign(n, 0);
    for (int c = 0; c < n; ++c) t.ok[c] = comp_ok[comp[c]];
    // a network that is not a tree may hold cells with two downstream rows: drop its pointers altogether
    for (int c = 0; c < n; ++c)
        if (!t.ok[c]) t.ds[c] = -1;
    t.nchild.assign(n, 0);
    for (int c = 0; c < n; ++c)
        if (t.ok[c] && t.ds[c] >= 0) t.nchild[t.ds[c]]++;
    t.child_ptr.assign(n + 1, 0);
    for (int c = 0; c < n; ++c) t.child_ptr[c + 1] = t.child_ptr[c] + t.nchild[c];
    t.child.assign(t.child_ptr[n], 0);
    {
        std::vector<int> fill(t.child_ptr.begin(), t.child_ptr.end() - 1);
        for (int c = 0; c < n; ++c)
            if (t.ok[c] && t.ds[c] >= 0) t.child[fill[t.ds[c]]++] = c;
    }
    // longest side of every row either side of its diagonal (the time-skewed kernels read that many values per sub-step)
    t.cell_pre.assign(n, 0);
    t.cell_post.assign(n, 0);
    for (int c = 0; c < n; ++c) {
        bool past = false;
        for (int64_t j = indptr[c]; j < indptr[c + 1]; ++j) {
            if (indices[j] == c) past = true;
            else ++(past ? t.cell_post[c] : t.cell_pre[c]);
        }
    }
}

namespace {

// Partition for one piece capacity: pieces, their stream edges and pipeline depth, units.
struct Partition {
    std::vector<int> queue, piece, closed_roots, piece_of_root, piece_size, piece_imp, piece_depth;
    std::vector<int> edge_prod_cell, edge_cons_cell, edge_of_prod;
    std::vector<int> unit_of_piece, unit_cells_n, unit_imp_n, unit_depth;
    int nunit = 0, nedge = 0, maxdepth = 0;
};

// values a unit reads per sub-step: chained when that saves a read (front side - one-by-one terms >= 2)
int reads_of(int pre, int dir, int post) { return std::max(pre - dir >= 2 ? dir + 1 : pre, 1) + std::max(post, 1); }

void make_partition(const Tree &t, const FlowPlanOptions &opt, int cap, Partition &P) {
    const int n = t.n;
    const int64_t *indptr = t.indptr;
    const int32_t *indices = t.indices;
    const std::vector<int> &ds = t.ds, &child = t.child, &child_ptr = t.child_ptr, &cell_pre = t.cell_pre,
                           &cell_post = t.cell_post;
    P = Partition();
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
    std::vector<int> open_cnt(n, 0), open_imp(n, 0);
    std::vector<int> &closed_roots = P.closed_roots;            // piece roots in closing order (upstream pieces first)
    std::vector<int> kids;
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        const int v = queue[qi];
        kids.assign(child.begin() + child_ptr[v], child.begin() + child_ptr[v + 1]);
        std::sort(kids.begin(), kids.end(), [&](int x, int y) {
            return open_cnt[x] != open_cnt[y] ? open_cnt[x] < open_cnt[y] : x < y;
        });
        int total = 1, imp = (int)kids.size();
        unsigned keep = 0;                          // bit i: kids[i]'s open piece joins v's
        for (size_t i = 0; i < kids.size(); ++i) {
            const int c = kids[i];
            if (total + open_cnt[c] <= cap && imp - 1 + open_imp[c] <= G_MAX) {
                keep |= 1u << i;
                total += open_cnt[c];
                imp += open_imp[c] - 1;
            }
        }
        if (opt.cut_rule && keep + 1 != (1u << kids.size()) && cell_pre[v] >= 3 && kids.size() <= 8) {
            // Not every child fits, and v's row has a long front side: which children become streams decides how many
            // values v reads per sub-step (a stream may only open the chain that sums the front side on the way, see
            // emit_tables).  Among the choices that fit: fewest reads for v, then most cells kept.
            auto v_reads = [&](unsigned kp) {
                int j = 0, k = 0;
                bool in_prefix = true;
                for (int64_t e = indptr[v]; e < indptr[v + 1]; ++e) {
                    const int src = indices[e];
                    if (src == v) break;
                    bool kept = false;
                    for (size_t i = 0; i < kids.size(); ++i)
                        if (kids[i] == src) kept = (kp >> i) & 1u;
                    if (in_prefix && (kept || k == 0)) ++j;
                    else in_prefix = false;
                    ++k;
                }
                const int dir = j >= 2 ? 1 + k - j : k;
                return (k - dir >= 2 ? dir + 1 : k) + cell_post[v];
            };
            // (the capacity is there for the packing; a piece around such a cell may grow up to a whole unit if that is
            // what it takes to keep its chain)
            int best_reads = v_reads(keep), best_total = total;
            for (unsigned kp = 0; kp < (1u << kids.size()); ++kp) {
                int tt = 1, im = (int)kids.size();
                for (size_t i = 0; i < kids.size(); ++i)
                    if ((kp >> i) & 1u) {
                        tt += open_cnt[kids[i]];
                        im += open_imp[kids[i]] - 1;
                    }
                if (tt > LANES || im > G_MAX) continue;
                const int r = v_reads(kp);
                const bool over = tt > cap, best_over = best_total > cap;
                if (r < best_reads || (r == best_reads && (over != best_over ? !over : tt > best_total))) {
                    best_reads = r;
                    best_total = tt;
                    keep = kp;
                }
            }
            total = 1;
            imp = (int)kids.size();
            for (size_t i = 0; i < kids.size(); ++i)
                if ((keep >> i) & 1u) {
                    total += open_cnt[kids[i]];
                    imp += open_imp[kids[i]] - 1;
                }
        }
        for (size_t i = 0; i < kids.size(); ++i) {
            const int c = kids[i];
            if ((keep >> i) & 1u) dsu[find(c)] = v;      // c's open piece joins v's
            else closed_roots.push_back(c);              // c's piece is final; its outlet streams into v
        }
        open_cnt[v] = total;
        open_imp[v] = imp;
        if (ds[v] < 0) {
            closed_roots.push_back(v);
        } else if (--left[ds[v]] == 0) {
            queue.push_back(ds[v]);
        }
    }

    // ---- pieces, their stream edges and pipeline depth
    const int npiece = (int)closed_roots.size();
    P.piece_of_root.assign(n, -1);
    for (int p = 0; p < npiece; ++p) P.piece_of_root[closed_roots[p]] = p;
    P.piece.assign(n, -1);
    P.piece_size.assign(npiece, 0);
    P.piece_imp.assign(npiece, 0);
    P.piece_depth.assign(npiece, 0);
    std::vector<int> ppre(npiece, 0), ppost(npiece, 0), pdir(npiece, 0);
    for (int v : queue) {
        const int q = P.piece_of_root[find(v)];
        P.piece[v] = q;
        P.piece_size[q]++;
        ppre[q] = std::max(ppre[q], cell_pre[v]);
        ppost[q] = std::max(ppost[q], cell_post[v]);
    }
    // front-side terms a cell still reads one by one when its unit is chained (see emit_tables): the prefix of cells of
    // its own piece (the first may be an imported stream) counts as one
    for (int c : queue) {
        const int q = P.piece[c];
        int j = 0, k = 0;
        bool in_prefix = true;
        for (int64_t e = indptr[c]; e < indptr[c + 1]; ++e) {
            const int src = indices[e];
            if (src == c) break;
            if (in_prefix && (P.piece[src] == q || k == 0)) ++j;
            else in_prefix = false;
            ++k;
        }
        pdir[q] = std::max(pdir[q], j >= 2 ? 1 + k - j : k);
    }
    P.edge_of_prod.assign(n, -1);                        // one stream per closed piece that has a downstream cell
    for (int p = 0; p < npiece; ++p) {                   // closing order: upstream pieces come first
        const int r = closed_roots[p];
        if (ds[r] >= 0) {
            const int cp = P.piece[ds[r]];
            P.edge_of_prod[r] = (int)P.edge_prod_cell.size();
            P.edge_prod_cell.push_back(r);
            P.edge_cons_cell.push_back(ds[r]);
            P.piece_imp[cp]++;
            P.piece_depth[cp] = std::max(P.piece_depth[cp], P.piece_depth[p] + 1);
        }
    }
    P.nedge = (int)P.edge_prod_cell.size();
    P.maxdepth = npiece ? *std::max_element(P.piece_depth.begin(), P.piece_depth.end()) : 0;

    // ---- packing.  Pieces with a stream in or out: equal depth per unit (a unit then only ever waits for units
    //      strictly upstream or downstream of it), first-fit decreasing.  Pieces
    //      without streams -- whole small networks -- wait for nobody and go wherever lanes are free; the `cheap_units`
    //      cheapest of them (fewest row terms) are kept together instead: units for the SIMDs that must hold two waves.
    auto has_out = [&](int p) { return ds[closed_roots[p]] >= 0; };
    auto terms_of = [&](int p) { return reads_of(ppre[p], pdir[p], ppost[p]); };
    std::vector<int> dep, fre;
    for (int p = 0; p < npiece; ++p) (P.piece_imp[p] > 0 || has_out(p) ? dep : fre).push_back(p);
    std::stable_sort(dep.begin(), dep.end(), [&](int x, int y) {
        if (P.piece_depth[x] != P.piece_depth[y]) return P.piece_depth[x] < P.piece_depth[y];
        return P.piece_size[x] > P.piece_size[y];
    });
    int cheap_units = 0;
    for (int round = 0; round < 4; ++round) {
        P.unit_of_piece.assign(npiece, -1);
        P.unit_cells_n.clear();
        P.unit_imp_n.clear();
        P.unit_depth.clear();
        std::vector<int> unit_out_n;                                   // outlets: <= G_MAX too
        std::vector<int> upre, udir, upost;                            // longest sides of the unit's rows
        auto new_unit = [&](int depth) {
            P.unit_cells_n.push_back(0);
            P.unit_imp_n.push_back(0);
            unit_out_n.push_back(0);
            upre.push_back(0);
            udir.push_back(0);
            upost.push_back(0);
            P.unit_depth.push_back(depth);
            return (int)P.unit_cells_n.size() - 1;
        };
        auto put_piece = [&](int p, int u) {
            P.unit_of_piece[p] = u;
            P.unit_cells_n[u] += P.piece_size[p];
            P.unit_imp_n[u] += P.piece_imp[p];
            unit_out_n[u] += has_out(p) ? 1 : 0;
            upre[u] = std::max(upre[u], ppre[p]);
            udir[u] = std::max(udir[u], pdir[p]);
            upost[u] = std::max(upost[u], ppost[p]);
        };
        // a piece joins a unit only if the unit then reads no more values per sub-step than the limit, or than the piece or
        // the unit need on their own: the slowest unit paces the run, and it is the one with the longest rows
        auto class_ok = [&](int p, int u) {
            const int tt = reads_of(std::max(upre[u], ppre[p]), std::max(udir[u], pdir[p]), std::max(upost[u], ppost[p]));
            return P.unit_cells_n[u] == 0 || tt <= std::max(opt.tlimit, std::max(terms_of(p), reads_of(upre[u], udir[u], upost[u])));
        };
        {
            size_t first_open = 0;
            int cur_depth = -1;
            for (int p : dep) {
                if (P.piece_depth[p] != cur_depth) {
                    cur_depth = P.piece_depth[p];
                    first_open = P.unit_cells_n.size();
                }
                int u = -1;
                for (size_t b = first_open; b < P.unit_cells_n.size(); ++b) {
                    if (P.unit_cells_n[b] + P.piece_size[p] <= LANES && P.unit_imp_n[b] + P.piece_imp[p] <= G_MAX &&
                        unit_out_n[b] + (has_out(p) ? 1 : 0) <= G_MAX && class_ok(p, (int)b)) {
                        u = (int)b;
                        break;
                    }
                }
                if (u < 0) u = new_unit(cur_depth);
                put_piece(p, u);
                while (first_open < P.unit_cells_n.size() && P.unit_cells_n[first_open] >= LANES) ++first_open;
            }
        }
        // the cheap units: free pieces by (row terms, size), filled one unit after the other
        std::vector<int> by_terms(fre);
        std::stable_sort(by_terms.begin(), by_terms.end(), [&](int x, int y) {
            return terms_of(x) != terms_of(y) ? terms_of(x) < terms_of(y) : P.piece_size[x] < P.piece_size[y];
        });
        std::vector<char> taken(npiece, 0);
        {
            int made = 0, u = -1;
            for (int p : by_terms) {
                if (terms_of(p) > 3) break;
                if (u < 0 || P.unit_cells_n[u] + P.piece_size[p] > LANES) {
                    if (made == cheap_units) break;
                    u = new_unit(0);
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
        std::stable_sort(by_size.begin(), by_size.end(), [&](int x, int y) { return P.piece_size[x] > P.piece_size[y]; });
        {
            // units by free lanes: bucket[f] = units with f free lanes
            std::vector<std::vector<int>> bucket(LANES + 1);
            for (int u = 0; u < (int)P.unit_cells_n.size(); ++u) bucket[LANES - P.unit_cells_n[u]].push_back(u);
            for (int p : by_size) {
                const int sz = P.piece_size[p];
                int u = -1;
                for (int pass = 0; pass < 2 && u < 0; ++pass)          // second pass: any unit with room
                    for (int f = sz; f <= LANES && u < 0; ++f)
                        for (size_t i = bucket[f].size(); i-- > 0;) {
                            const int b = bucket[f][i];
                            if (pass == 1 || class_ok(p, b)) {
                                u = b;
                                bucket[f].erase(bucket[f].begin() + (long)i);
                                break;
                            }
                        }
                if (u < 0) u = new_unit(0);
                put_piece(p, u);
                bucket[LANES - P.unit_cells_n[u]].push_back(u);
            }
        }
        P.nunit = (int)P.unit_cells_n.size();
        const int need = opt.simds > 0 ? std::max(P.nunit - opt.simds, 0) : 0;
        if (need <= cheap_units) break;
        cheap_units = need + (round > 0 ? 2 : 0);      // the cheap units themselves may add a unit or two
    }
}

}  // namespace

// ---- lanes against LDS bank conflicts ---------------------------------------------------------------------------------
// A unit's gather is one ds_read_b128 per row term, each lane addressing the LDS entry of one upstream neighbour.  The LDS
// serves a wave's read in fixed lane groups -- four of 16 lanes for b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the
// same + 32) -- one cycle per group when the lanes of a group address distinct 16-byte columns of the 256-byte bank row; every further distinct address on a busy
// column costs the whole CU one more LDS cycle (identical addresses are one broadcast).  Cells were laid on the lanes in
// the order of their ids: neighbours sit near each other and most reads are clean, but a sixth of the LDS cycles of a run
// were conflicts (SQ_LDS_BANK_CONFLICT).  Which lane holds which cell is free -- every table goes through the
// slot -- so the cells of a unit (and, among themselves, its imported entries) are moved to lanes on which the unit's
// reads collide least: a few dozen directed swaps per unit, each kept if the count of extra LDS cycles does not rise.
struct LaneOpt {
    const FlowTables &t;
    int u;
    int64_t ts;
    bool chained;
    int pre, post, ng;
    int pe[NPAIR];                       // entry (old) -> entry (new); lanes 0..63, imported entries 64..127, zero 128
    std::vector<int> offenders;          // old entries that sit on a busy column as the second, third .. address

    static int group128(int lane) { return ((lane >> 5) << 1) | (int)((0x0f0ff0f0u >> (lane & 31)) & 1u); }

    LaneOpt(const FlowTables &tt, int unit, int nghost) : t(tt), u(unit), ts((int64_t)tt.n_units * LANES), ng(nghost) {
        chained = (t.unit_p[u] & 0x100) != 0;
        pre = t.unit_p[u] & 15;
        post = (t.unit_p[u] >> 4) & 15;
        std::iota(pe, pe + NPAIR, 0);
    }
    // extra LDS cycles per sub-step of the unit's gather under `pe`
    int cost(bool want_offenders) {
        if (want_offenders) offenders.clear();
        constexpr int ngroup = 4, ncol = 16;
        int total = 0;
        int seen[4][16][8], who[4][16][8], nseen[4][16];      // distinct addresses per (group, column) and a lane reading each; 8 hold any realistic pile-up
        auto column = [&](const unsigned *tab) {
            for (int g = 0; g < ngroup; ++g)
                for (int c = 0; c < ncol; ++c) nseen[g][c] = 0;
            for (int l = 0; l < LANES; ++l) {
                const int e_old = (int)(tab[(int64_t)u * LANES + l] >> 4), e = pe[e_old];
                const int lane = pe[l], g = group128(lane), c = e & (ncol - 1);
                int k = 0, &m = nseen[g][c];
                while (k < m && k < 8 && seen[g][c][k] != e) ++k;
                if (k < m || k >= 8) continue;                   // the same address again: a broadcast
                seen[g][c][m] = e;
                who[g][c][m++] = l;
            }
            for (int g = 0; g < ngroup; ++g) {
                int worst = 1;
                for (int c = 0; c < ncol; ++c) worst = std::max(worst, nseen[g][c]);
                total += worst - 1;
                if (!want_offenders || worst == 1) continue;
                for (int c = 0; c < ncol; ++c)                   // everybody on a worst column: the entries read and the reading lanes
                    if (nseen[g][c] == worst)
                        for (int k = 0; k < worst; ++k) {
                            const int e_old = (int)(tab[(int64_t)u * LANES + who[g][c][k]] >> 4);
                            if (e_old != 2 * LANES) offenders.push_back(e_old);
                            offenders.push_back(who[g][c][k]);
                        }
            }
        };
        for (int w = 0; w < pre && w < SK_P; ++w) column(t.ent2.data() + (size_t)w * ts);
        for (int w = 0; w < post && w < SK_P; ++w) column(t.ent2.data() + (size_t)(SK_P + w) * ts);
        if (chained) column(t.eprev.data());
        return total;
    }
    // returns (extra cycles before, after)
    std::pair<int, int> run(int trials) {
        const int before = cost(true);
        int cur = before;
        unsigned long long rng = 0x9e3779b97f4a7c15ull * (unsigned long long)(u + 1);
        auto next = [&]() {
            rng ^= rng << 13;
            rng ^= rng >> 7;
            rng ^= rng << 17;
            return (unsigned)(rng >> 11);
        };
        for (int it = 0; it < trials && cur > 0 && !offenders.empty(); ++it) {
            const int a = offenders[next() % offenders.size()];
            int b;
            if (a < LANES) b = (int)(next() % LANES);
            else if (ng > 1) b = LANES + (int)(next() % (unsigned)ng);
            else continue;
            if (a == b) continue;
            std::swap(pe[a], pe[b]);
            const std::vector<int> keep = offenders;
            const int c2 = cost(true);
            if (c2 <= cur) cur = c2;
            else {
                std::swap(pe[a], pe[b]);
                offenders = keep;
            }
        }
        return {before, cur};
    }
};

// moves every per-slot table of unit u to the lanes LaneOpt chose
void lane_apply(FlowTables &t, int u, const int *pe) {
    const int64_t ts = (int64_t)t.n_units * LANES, base = (int64_t)u * LANES;
    auto move_rows = [&](auto &vec, int64_t off, int lo, int hi) {        // rows lo..hi-1 of the unit go to pe[row] - lo
        typename std::remove_reference<decltype(vec)>::type tmp(vec.begin() + off + base, vec.begin() + off + base + LANES);
        for (int l = lo; l < hi; ++l) vec[off + base + (pe[lo == 0 ? l : LANES + l] - (lo == 0 ? 0 : LANES))] = tmp[l];
    };
    auto remap = [&](unsigned &o) {                                      // an entry offset: entry x 16 + low bits
        const unsigned e = o >> 4;
        if (e < (unsigned)NPAIR) o = ((unsigned)pe[e] << 4) | (o & 15u);
    };
    for (int64_t w = 0; w < W_MAX; ++w)
        for (int l = 0; l < LANES; ++l) remap(t.ent[(size_t)(w * ts + base + l)]);
    for (int64_t w = 0; w < 2 * SK_P; ++w)
        for (int l = 0; l < LANES; ++l) remap(t.ent2[(size_t)(w * ts + base + l)]);
    for (int l = 0; l < LANES; ++l) remap(t.eprev[(size_t)(base + l)]);
    move_rows(t.cell_of_slot, 0, 0, LANES);
    move_rows(t.export_edge, 0, 0, LANES);
    move_rows(t.lag, 0, 0, LANES);
    move_rows(t.eprev, 0, 0, LANES);
    move_rows(t.lane_flags, 0, 0, LANES);
    for (int64_t w = 0; w < W_MAX; ++w) move_rows(t.ent, w * ts, 0, LANES);
    for (int64_t w = 0; w < 2 * SK_P; ++w) move_rows(t.ent2, w * ts, 0, LANES);
    // imported entries: ghost k of the unit is row k of these
    std::vector<int> ge(t.ghost_edge.begin() + base, t.ghost_edge.begin() + base + LANES),
        gl(t.ghost_lag.begin() + base, t.ghost_lag.begin() + base + LANES),
        gp(t.ghost_prod.begin() + base, t.ghost_prod.begin() + base + LANES);
    for (int k = 0; k < LANES; ++k) {
        const int k2 = pe[LANES + k] - LANES;
        t.ghost_edge[base + k2] = ge[k];
        t.ghost_lag[base + k2] = gl[k];
        t.ghost_prod[base + k2] = gp[k];
    }
}

void lane_optimise(FlowTables &t, int trials, bool debug) {
    const int nunit = t.n_units;
    std::vector<int> before(nunit, 0), after(nunit, 0);
    std::vector<int> perm((size_t)nunit * NPAIR);
    const int nthread = std::max(1, std::min(8, (int)std::thread::hardware_concurrency()));
    auto work = [&](int k) {
        for (int u = k; u < nunit; u += nthread) {
            int ng = 0;
            while (ng < LANES && t.ghost_edge[(int64_t)u * LANES + ng] >= 0) ++ng;
            LaneOpt lo(t, u, ng);
            const auto r = lo.run(trials);
            before[u] = r.first;
            after[u] = r.second;
            std::copy(lo.pe, lo.pe + NPAIR, perm.begin() + (size_t)u * NPAIR);
        }
    };
    std::vector<std::thread> th;
    for (int k = 1; k < nthread; ++k) th.emplace_back(work, k);
    work(0);
    for (auto &x : th) x.join();
    int64_t sb = 0, sa = 0, reads = 0;
    for (int u = 0; u < nunit; ++u) {
        if (after[u] < before[u]) lane_apply(t, u, perm.data() + (size_t)u * NPAIR);
        sb += before[u];
        sa += std::min(after[u], before[u]);
        reads += (t.unit_p[u] & 15) + ((t.unit_p[u] >> 4) & 15) + ((t.unit_p[u] & 0x100) ? 1 : 0);
    }
    if (debug)
        fprintf(stderr, "  lanes: %lld extra LDS cycles per sub-step over %d units as laid out by cell id, %lld after %d swaps tried per unit (%lld reads)\n",
                (long long)sb, nunit, (long long)sa, trials, (long long)reads);
}

int flow_tables_build(int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign, const int *comp,
                      int ncomp, const FlowPlanOptions &opt, std::vector<char> &handled, FlowTables &out,
                      std::string &err) {
    out = FlowTables();
    handled.assign(n, 0);
    if (n == 0) return 0;
    Tree t;
    tree_analyse(n, indptr, indices, sign, comp, ncomp, t);
    const std::vector<int> &ds = t.ds;

    // ---- the piece capacity.  A smaller capacity than LANES costs streams (every cut is one) and buys units: pieces of
    //      33..64 cells cannot share a unit, so the largest capacity leaves every other unit ~10 lanes short of full
    //      (67,420 cells: 1,121 units at 64; 1,059 at 36 -- 1,054 would be every lane used).  Units beyond the SIMD count
    //      share a SIMD with another unit and the slowest unit paces the run, so units are what counts.
    Partition P;
    {
        static const int caps[] = {LANES, 56, 48, 44, 40, 36, 32};
        constexpr int NCAP = (int)(sizeof(caps) / sizeof(caps[0]));
        // the candidate partitions do not depend on each other: one host thread each (67,420 cells: 7 x ~10 ms otherwise,
        // inside run_model()'s wall time); the choice below walks them in the order of the list as before
        std::vector<Partition> cand(opt.piece_cap > 0 ? 1 : NCAP);
        if (cand.size() == 1 || n < 4096) {
            for (size_t k = 0; k < cand.size(); ++k)
                make_partition(t, opt, opt.piece_cap > 0 ? std::min(opt.piece_cap, LANES) : caps[k], cand[k]);
        } else {
            std::vector<std::thread> pool;
            for (size_t k = 0; k < cand.size(); ++k)
                pool.emplace_back([&, k] { make_partition(t, opt, caps[k], cand[k]); });
            for (auto &th : pool) th.join();
        }
        bool have = false;
        long best_score = 0;
        for (size_t k = 0; k < cand.size(); ++k) {
            Partition &Q = cand[k];
            const int cap = opt.piece_cap > 0 ? std::min(opt.piece_cap, LANES) : caps[k];
            // units beyond the SIMD count share a SIMD; only units without streams may (see the claim order below): a unit
            // with streams that has to share one slows every unit it is linked to, which costs far more than a few units
            int indep = 0;
            {
                std::vector<char> coupled(Q.nunit, 0);
                for (size_t p = 0; p < Q.closed_roots.size(); ++p)
                    if (Q.piece_imp[p] > 0 || ds[Q.closed_roots[p]] >= 0) coupled[Q.unit_of_piece[p]] = 1;
                for (int u = 0; u < Q.nunit; ++u) indep += coupled[u] ? 0 : 1;
            }
            const int extra = opt.simds > 0 ? std::max(Q.nunit - opt.simds, 0) : 0;
            const long score = 1000000L * std::max(2 * extra - indep, 0) + 1000L * Q.nunit + Q.nedge / 8;
            if (opt.debug)
                fprintf(stderr, "flow plan: piece capacity %d -> %d units, %d streams, %d units without streams\n", cap,
                        Q.nunit, Q.nedge, indep);
            if (!have || score < best_score) {
                std::swap(P, Q);
                best_score = score;
                have = true;
            }
            if (opt.simds > 0 && P.nunit <= opt.simds) break;       // every unit has a SIMD of its own
        }
    }
    const std::vector<int> &queue = P.queue, &piece = P.piece, &closed_roots = P.closed_roots, &piece_of_root = P.piece_of_root;
    const std::vector<int> &piece_depth = P.piece_depth;
    const std::vector<int> &edge_prod_cell = P.edge_prod_cell, &edge_cons_cell = P.edge_cons_cell, &edge_of_prod = P.edge_of_prod;
    const std::vector<int> &unit_of_piece = P.unit_of_piece, &unit_imp_n = P.unit_imp_n;
    const int npiece = (int)closed_roots.size();
    const int nedge = P.nedge, maxdepth = P.maxdepth;
    const int nunit = P.nunit;
    if (nunit == 0) return 0;

    // ---- slots, ghosts, gather offsets
    const int64_t ts = (int64_t)nunit * LANES;
    std::vector<int> &cell_of_slot = out.cell_of_slot, &export_edge = out.export_edge, &ghost_edge = out.ghost_edge;
    cell_of_slot.assign(ts, -1);
    export_edge.assign(ts, -1);
    ghost_edge.assign(ts, -1);
    std::vector<int> slot_of_cell(n, -1);
    std::vector<int> fill(nunit, 0), gfill(nunit, 0), edge_ghost(nedge);
    std::vector<int> &edge_cons_unit = out.edge_cons_unit;
    edge_cons_unit.assign(nedge, 0);
    for (int c = 0; c < n; ++c)
        if (piece[c] >= 0) {
            const int u = unit_of_piece[piece[c]];
            const int s = fill[u]++;
            if (s >= LANES) {
                err = "flow plan: a unit holds more than 64 cells";
                return -1;
            }
            cell_of_slot[(int64_t)u * LANES + s] = c;
            slot_of_cell[c] = s;
            handled[c] = 1;
        }
    for (int ed = 0; ed < nedge; ++ed) {
        const int u = unit_of_piece[piece[edge_cons_cell[ed]]];
        const int g = gfill[u]++;
        if (g >= LANES) {
            err = "flow plan: a unit imports more than 64 streams";
            return -1;
        }
        edge_cons_unit[ed] = u;
        edge_ghost[ed] = g;
        ghost_edge[(int64_t)u * LANES + g] = ed;
        const int pc = edge_prod_cell[ed];
        export_edge[(int64_t)unit_of_piece[piece[pc]] * LANES + slot_of_cell[pc]] = ed;
    }
    out.ent.assign((size_t)W_MAX * ts, (unsigned)(NPAIR - 1) * 16u);
    out.unit_terms.assign(nunit, 1);
    for (int c = 0; c < n; ++c) {
        if (piece[c] < 0) continue;
        const int u = unit_of_piece[piece[c]];
        out.unit_terms[u] = std::max(out.unit_terms[u], (int)(indptr[c + 1] - indptr[c]));
        const int64_t slot = (int64_t)u * LANES + slot_of_cell[c];
        int w = 0;
        for (int64_t j = indptr[c]; j < indptr[c + 1]; ++j, ++w) {
            const int src = indices[j];
            unsigned off;
            if (piece[src] >= 0 && unit_of_piece[piece[src]] == u) {
                off = (unsigned)slot_of_cell[src] * 16u + (sign[j] < 0 ? 8u : 0u);
            } else {                                    // the outlet of an upstream piece in another unit
                const int ed = edge_of_prod[src];
                if (ed < 0 || edge_cons_unit[ed] != u) {
                    err = "flow plan: inconsistent stream edge at cell " + std::to_string(c);
                    return -1;
                }
                off = (unsigned)(LANES + edge_ghost[ed]) * 16u;
            }
            out.ent[(size_t)w * ts + slot] = off;
        }
    }

    // ---- time-skewed layout.  Lane lags: a cell `h` edges above its piece's outlet runs 2 * (H - h) sub-steps behind the
    //      unit's clock (H = tallest piece of the unit, an imported stream counting as one more level), so that every
    //      flow a cell gathers was produced exactly two iterations earlier.  The lags of a unit are shifted so that its
    //      outlets' lag is a multiple of 16 (stream stores of 16 sub-steps never wrap inside a group).  Row terms are
    //      split at the diagonal: SK_P before, SK_P after.
    bool skew_ok = true;
    // Chained units: the cells that feed a cell from in front of its diagonal, as far as they are lanes of the unit from
    // the first one on (an imported stream ends the chain: its value is dropped into LDS by the block transfers, not
    // computed by a lane), pass a running sum along their stored order; the fed cell reads the last one's value as one
    // term and the rest of its front side term by term.  A unit is chained when that saves at least one read per
    // sub-step: longest front side (reads saved + 1 for the running value) at least 2 shorter.
    std::vector<char> unit_chain(nunit, 0);
    std::vector<int> chain_extra(n, 0), chain_prev(n, -1), chain_len(n, 0);   // levels below the last of the chain; cell before
    std::vector<int> edge_reader(edge_cons_cell);      // the cell whose lane reads an imported value out of LDS
    std::vector<int> upre(nunit, 0), udir(nunit, 0);
    {
        auto front = [&](int c, int u, int &k) {       // k = terms in front of the diagonal, returns the chainable prefix:
            int j = 0;                                 // lanes of the unit, the first one possibly an imported stream
            bool in_prefix = true;
            k = 0;
            for (int64_t e = indptr[c]; e < indptr[c + 1]; ++e) {
                const int src = indices[e];
                if (src == c) break;
                const bool inu = piece[src] >= 0 && unit_of_piece[piece[src]] == u;
                if (in_prefix && (inu || k == 0)) ++j;
                else in_prefix = false;
                ++k;
            }
            return j;
        };
        for (int c = 0; c < n; ++c) {
            if (piece[c] < 0) continue;
            const int u = unit_of_piece[piece[c]];
            int k;
            const int j = front(c, u, k);
            upre[u] = std::max(upre[u], k);
            udir[u] = std::max(udir[u], j >= 2 ? 1 + k - j : k);
        }
        for (int u = 0; u < nunit; ++u) unit_chain[u] = opt.chain && upre[u] - udir[u] >= 2 && udir[u] <= 2;
        for (int c = 0; c < n; ++c) {
            if (piece[c] < 0) continue;
            const int u = unit_of_piece[piece[c]];
            if (!unit_chain[u]) continue;
            int k;
            const int j = front(c, u, k);
            if (j < 2) continue;
            chain_len[c] = j;
            int prev = -1;
            for (int i = 0; i < j; ++i) {
                const int tc = indices[indptr[c] + i];
                const bool inu = piece[tc] >= 0 && unit_of_piece[piece[tc]] == u;
                if (!inu) {             // an imported stream opens the chain: the next cell adds its flows to the ghost value
                    edge_reader[edge_of_prod[tc]] = indices[indptr[c] + 1];
                    prev = -2 - edge_of_prod[tc];
                    continue;
                }
                chain_extra[tc] = j - 1 - i;
                chain_prev[tc] = prev;
                prev = tc;
            }
        }
    }
    std::vector<int> hgt(n, 0), unit_h(nunit, 0);
    for (size_t qi = queue.size(); qi-- > 0;) {          // reverse bottom-up order: downstream cells first
        const int c = queue[qi];
        if (piece[c] < 0) continue;
        hgt[c] = (piece_of_root[c] == piece[c]) ? 0 : hgt[ds[c]] + 1 + chain_extra[c];
        int &uh = unit_h[unit_of_piece[piece[c]]];
        uh = std::max(uh, hgt[c]);
    }
    for (int ed = 0; ed < nedge; ++ed) {
        int &uh = unit_h[edge_cons_unit[ed]];
        uh = std::max(uh, hgt[edge_reader[ed]] + 1);
    }
    out.lag.assign(ts, 0);
    out.ghost_lag.assign(ts, 0);
    out.unit_p.assign(nunit, 0x11);
    out.unit_lmax.assign(nunit, 0);
    out.unit_glmax.assign(nunit, 0);
    out.ent2.assign((size_t)2 * SK_P * ts, SK_ZERO);
    out.eprev.assign(ts, SK_ZERO);
    for (int u = 0; u < nunit; ++u) out.unit_lmax[u] = (2 * unit_h[u] + 15) & ~15;
    for (int c = 0; c < n; ++c) {
        if (piece[c] < 0) continue;
        const int u = unit_of_piece[piece[c]];
        const int64_t slot = (int64_t)u * LANES + slot_of_cell[c];
        out.lag[slot] = out.unit_lmax[u] - 2 * hgt[c];
        if (chain_prev[c] >= 0) out.eprev[slot] = (unsigned)slot_of_cell[chain_prev[c]] * 16u;
        else if (chain_prev[c] <= -2) out.eprev[slot] = (unsigned)(LANES + edge_ghost[-2 - chain_prev[c]]) * 16u;
        int npre = 0, npost = 0, seen = 0;
        bool past = false;
        for (int64_t j = indptr[c]; j < indptr[c + 1]; ++j) {
            const int src = indices[j];
            if (src == c) {
                past = true;
                continue;
            }
            if (!past && ++seen < chain_len[c]) continue;      // summed on the way: only the last of the chain is read
            unsigned off;
            if (piece[src] >= 0 && unit_of_piece[piece[src]] == u) off = (unsigned)slot_of_cell[src] * 16u;
            else off = (unsigned)(LANES + edge_ghost[edge_of_prod[src]]) * 16u;
            int &k = past ? npost : npre;
            if (k >= SK_P) {
                skew_ok = false;
                continue;
            }
            out.ent2[(size_t)((past ? SK_P : 0) + k) * ts + slot] = off;
            ++k;
        }
        int &up = out.unit_p[u];
        up = std::max(up & 15, npre) | (std::max((up >> 4) & 15, npost) << 4) | (unit_chain[u] ? 0x100 : 0);
    }
    for (int ed = 0; ed < nedge; ++ed) {
        const int u = edge_cons_unit[ed];
        const int gl = out.unit_lmax[u] - 2 * (hgt[edge_reader[ed]] + 1);
        out.ghost_lag[(int64_t)u * LANES + edge_ghost[ed]] = gl;
        out.unit_glmax[u] = std::max(out.unit_glmax[u], gl);
    }

    out.lane_flags.assign(ts, 0);      // (the reassociated planner's: xh_flow_rsum.cpp)
    out.ghost_prod.assign(ts, 0);
    for (int ed = 0; ed < nedge; ++ed) out.ghost_prod[(int64_t)edge_cons_unit[ed] * LANES + edge_ghost[ed]] = edge_prod_cell[ed];

    std::vector<int> unit_exp(nunit, 0);
    for (int ed = 0; ed < nedge; ++ed) unit_exp[unit_of_piece[piece[edge_prod_cell[ed]]]]++;
    {   // ---- the claim list.  With more units than SIMDs some SIMDs hold two waves; two waves on a SIMD take about as long
        //      as their instruction streams put together.  A unit with streams passes its delay on to every unit linked to
        //      it; a unit without streams only delays itself.  So the SIMDs with two waves should hold units without
        //      streams, a cheap one next to a dearer one that gets issue priority.  Which workgroup lands on which SIMD
        //      cannot be planned (it follows the workgroup id only on an idle device), so every workgroup finds out where
        //      it runs and claims its unit from this list (top of k_mrtm_wave): units without streams by rising cost, then
        //      the others.
        std::vector<int> cost(nunit);
        for (int u = 0; u < nunit; ++u) {
            const int reads = (out.unit_p[u] & 15) + ((out.unit_p[u] >> 4) & 15) + ((out.unit_p[u] & 0x100) ? 1 : 0);
            // measured per sub-step (profiles/round3): 138 + 25 per value read
            cost[u] = 138 + 25 * reads + (unit_imp_n[u] > 0 ? 15 : 0) + (unit_exp[u] > 0 ? 15 : 0);
        }
        auto coupled = [&](int u) { return unit_imp_n[u] > 0 || unit_exp[u] > 0; };
        out.unit_order.resize(nunit);
        std::iota(out.unit_order.begin(), out.unit_order.end(), 0);
        std::stable_sort(out.unit_order.begin(), out.unit_order.end(), [&](int x, int y) {
            return coupled(x) != coupled(y) ? !coupled(x) : cost[x] < cost[y];
        });
        // XH_WAVE_BALANCE=1 (experiment, round 4; off by default): behind the entries that go to the SIMDs with two waves
        // (2 x (units - SIMDs) of them) the list is ordered by what a unit asks of its CU's LDS per sub-step (values read x
        // their size), and the kernel hands the k-th quarter of this tail to the workgroups on SIMD k of their CU, so that
        // every CU holds one unit of each quarter and no CU four heavy ones (a unit loses 10 - 12 % to its CU mates,
        // profiles/round3/pmc_cu_sharing.txt).  Measured at the full grid, same box, twice: 23.53 / 23.57 ms in arrival
        // order, 23.64 / 23.65 ms balanced -- no gain: what a unit loses to its mates does not follow their LDS bytes.
        if (opt.simds > 0 && opt.balance_lds) {
            const int two = std::min(nunit, 2 * std::max(nunit - opt.simds, 0));
            auto lds_load = [&](int u) {
                const int reads = (out.unit_p[u] & 15) + ((out.unit_p[u] >> 4) & 15) + ((out.unit_p[u] & 0x100) ? 1 : 0);
                return reads * 16;
            };
            std::stable_sort(out.unit_order.begin() + two, out.unit_order.end(),
                             [&](int x, int y) { return lds_load(x) < lds_load(y); });
        }
    }

    out.skew_ok = skew_ok;
    out.skew_lmax = *std::max_element(out.unit_lmax.begin(), out.unit_lmax.end());
    {   // longest jump of a stream over pipeline levels: the ring of such a stream has to hold what the levels in between
        // need as lead (wave_launch, ring size)
        int span = 1;
        for (int ed = 0; ed < nedge; ++ed)
            span = std::max(span, piece_depth[piece[edge_cons_cell[ed]]] - piece_depth[piece[edge_prod_cell[ed]]]);
        out.skew_span = span;
    }
    out.n_units = nunit;
    out.n_edges = nedge;
    out.depth = maxdepth + 1;
    out.n_cells = (int)std::count(handled.begin(), handled.end(), (char)1);
    out.max_imports = *std::max_element(unit_imp_n.begin(), unit_imp_n.end());
    out.max_exports = *std::max_element(unit_exp.begin(), unit_exp.end());
    out.edge_prod_cell = edge_prod_cell;
    out.edge_cons_cell = edge_cons_cell;
    out.unit_depth = P.unit_depth;
    out.piece_of_cell = piece;
    out.unit_of_cell.assign(n, -1);
    for (int c = 0; c < n; ++c)
        if (piece[c] >= 0) out.unit_of_cell[c] = unit_of_piece[piece[c]];
    out.height_of_cell = hgt;
    out.ds = ds;
    if (opt.lane_trials > 0) lane_optimise(out, opt.lane_trials, opt.debug);

    if (opt.debug) {      // partition statistics on stderr
        std::vector<int> hp(8, 0), hi(9, 0), hx(9, 0), hl(10, 0), hpp(25, 0);
        int n_chain = 0;
        auto bucket = [](int v) { return v == 0 ? 0 : v <= 1 ? 1 : v <= 2 ? 2 : v <= 4 ? 3 : v <= 8 ? 4 : v <= 16 ? 5 : v <= 32 ? 6 : 7; };
        for (int u = 0; u < nunit; ++u) {
            hp[std::max(out.unit_p[u] & 15, (out.unit_p[u] >> 4) & 15)]++;
            hpp[(out.unit_p[u] & 15) * 5 + ((out.unit_p[u] >> 4) & 15)]++;
            if (out.unit_p[u] & 0x100) ++n_chain;
            hi[bucket(unit_imp_n[u])]++;
            hx[bucket(unit_exp[u])]++;
            hl[std::min(out.unit_lmax[u] / 16, 9)]++;
        }
        fprintf(stderr, "flow plan: %d units, %d pieces, %d edges, depth %d, skew_ok %d\n", nunit, npiece, nedge, maxdepth + 1,
                (int)skew_ok);
        fprintf(stderr, "  chained units: %d\n", n_chain);
        fprintf(stderr, "  units by P (1..4):");
        for (int k = 1; k <= 4; ++k) fprintf(stderr, " %d", hp[k]);
        fprintf(stderr, "\n  units by (pre, post) terms:");
        for (int a = 1; a <= 4; ++a)
            for (int b = 1; b <= 4; ++b)
                if (hpp[a * 5 + b]) fprintf(stderr, " (%d,%d) %d", a, b, hpp[a * 5 + b]);
        fprintf(stderr, "\n  units by imports (0,1,2,<=4,<=8,<=16,<=32,more):");
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %d", hi[k]);
        fprintf(stderr, "\n  units by exports (0,1,2,<=4,<=8,<=16,<=32,more):");
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %d", hx[k]);
        fprintf(stderr, "\n  units by lmax/16 (0..9+):");
        for (int k = 0; k < 10; ++k) fprintf(stderr, " %d", hl[k]);
        fprintf(stderr, "\n  longest stream jump: %d levels\n", out.skew_span);
    }
    return 0;
}

std::string flow_tables_check(int n, const int64_t *indptr, const int32_t *indices, const int8_t *sign,
                              const std::vector<char> &handled, const FlowTables &t) {
    (void)sign;
    auto fail = [](const std::string &s) { return s; };
    if (t.n_units == 0) {
        for (int c = 0; c < n; ++c)
            if (handled[c]) return fail("handled cell without units");
        return "";
    }
    const int64_t ts = (int64_t)t.n_units * LANES;
    if ((int64_t)t.cell_of_slot.size() != ts || (int64_t)t.lag.size() != ts || (int64_t)t.eprev.size() != ts ||
        (int64_t)t.ent2.size() != 2 * SK_P * ts || (int)t.unit_p.size() != t.n_units)
        return fail("table sizes");
    // every handled cell in exactly one slot
    std::vector<int> slot_of(n, -1);
    for (int64_t s = 0; s < ts; ++s) {
        const int c = t.cell_of_slot[s];
        if (c < 0) continue;
        if (c >= n || !handled[c]) return fail("slot holds a cell that is not handled");
        if (slot_of[c] >= 0) return fail("cell " + std::to_string(c) + " sits in two slots");
        slot_of[c] = (int)s;
    }
    for (int c = 0; c < n; ++c)
        if (handled[c] && slot_of[c] < 0) return fail("handled cell " + std::to_string(c) + " has no slot");
    // streams: producer / consumer bookkeeping, limits, direction
    std::vector<int> imp(t.n_units, 0), exp(t.n_units, 0);
    for (int ed = 0; ed < t.n_edges; ++ed) {
        const int pc = t.edge_prod_cell[ed], cc = t.edge_cons_cell[ed];
        if (t.ds[pc] != cc) return fail("stream edge does not follow a flow edge");
        const int pu = slot_of[pc] / LANES, cu = slot_of[cc] / LANES;
        if (t.export_edge[slot_of[pc]] != ed) return fail("export_edge of the producer");
        if (t.edge_cons_unit[ed] != cu) return fail("edge_cons_unit");
        if (pu == cu && t.piece_of_cell[pc] == t.piece_of_cell[cc]) return fail("stream inside a piece");
        if (t.unit_depth[pu] >= t.unit_depth[cu] && !(t.unit_depth[pu] == 0 && t.unit_depth[cu] == 0 && false))
            return fail("stream from depth " + std::to_string(t.unit_depth[pu]) + " to depth " + std::to_string(t.unit_depth[cu]));
        imp[cu]++;
        exp[pu]++;
    }
    for (int u = 0; u < t.n_units; ++u) {
        if (imp[u] > G_MAX || exp[u] > G_MAX) return fail("more than 16 imports / outlets in unit " + std::to_string(u));
        int g = 0;
        for (int k = 0; k < LANES; ++k)
            if (t.ghost_edge[(int64_t)u * LANES + k] >= 0) {
                if (k != g) return fail("ghost entries are not packed from 0");
                ++g;
            }
        if (g != imp[u]) return fail("ghost count");
        if (t.unit_lmax[u] & 15) return fail("unit lag not a multiple of 16");
    }
    // lags, and every row re-derived from the tables: a value is either a cell's own flow, an imported stream, or --
    // for a chain member -- the running sum of the chain up to and including it
    auto expand = [&](int u, unsigned off, std::vector<int> &terms, int depth, auto &&self) -> bool {
        if (off == SK_ZERO) return true;
        const int e = (int)(off / 16u);
        if (off % 16u) return false;
        if (e >= LANES) {                                   // ghost
            const int ed = t.ghost_edge[(int64_t)u * LANES + (e - LANES)];
            if (ed < 0) return false;
            terms.push_back(t.edge_prod_cell[ed]);
            return true;
        }
        const int64_t s = (int64_t)u * LANES + e;
        const int c = t.cell_of_slot[s];
        if (c < 0 || depth > LANES) return false;
        if ((t.unit_p[u] & 0x100) && t.eprev[s] != SK_ZERO && !self(u, t.eprev[s], terms, depth + 1, self)) return false;
        terms.push_back(c);
        return true;
    };
    auto lag_of = [&](int u, unsigned off) {                // lag of the lane / ghost that produces the value
        const int e = (int)(off / 16u);
        return e >= LANES ? t.ghost_lag[(int64_t)u * LANES + (e - LANES)] : t.lag[(int64_t)u * LANES + e];
    };
    std::vector<int> terms, want;
    for (int c = 0; c < n; ++c) {
        if (!handled[c]) continue;
        const int s = slot_of[c], u = s / LANES;
        if (t.lag[s] < 0 || t.lag[s] > t.unit_lmax[u] || (t.lag[s] & 1)) return fail("lag of cell " + std::to_string(c));
        const bool chained = (t.unit_p[u] & 0x100) != 0;
        if (chained && t.eprev[s] != SK_ZERO && t.lag[s] - lag_of(u, t.eprev[s]) != 2) return fail("chain lag");
        if (!chained && t.eprev[s] != SK_ZERO) return fail("eprev in a unit that is not chained");
        const int upre = t.unit_p[u] & 15, upost = (t.unit_p[u] >> 4) & 15;
        for (int side = 0; side < 2; ++side) {
            terms.clear();
            want.clear();
            bool past = false;
            for (int64_t j = indptr[c]; j < indptr[c + 1]; ++j) {
                if (indices[j] == c) past = true;
                else if (past == (side == 1)) want.push_back(indices[j]);
            }
            for (int w = 0; w < SK_P; ++w) {
                const unsigned off = t.ent2[(size_t)(side * SK_P + w) * ts + s];
                if (off == SK_ZERO) continue;
                if (w >= (side ? upost : upre)) return fail("row longer than the unit's shape at cell " + std::to_string(c));
                if (t.lag[s] - lag_of(u, off) != 2) return fail("a gathered value is not two iterations old at cell " + std::to_string(c));
                if (side == 1) {                            // behind the diagonal nothing is summed on the way
                    const int e = (int)(off / 16u);
                    if (e < LANES && chained && t.eprev[(int64_t)u * LANES + e] != SK_ZERO) return fail("chain member read from behind a diagonal");
                }
                if (!expand(u, off, terms, 0, expand)) return fail("bad table entry at cell " + std::to_string(c));
            }
            if (terms != want) return fail("row of cell " + std::to_string(c) + " does not expand to its CSR row");
        }
        // a chain member's running value may only be read by the next member or by the cell the chain feeds
        if (chained && t.eprev[s] != SK_ZERO && t.ds[c] < 0) return fail("chain member without a downstream cell");
    }
    for (int ed = 0; ed < t.n_edges; ++ed) {
        const int cu = slot_of[t.edge_cons_cell[ed]] / LANES;
        bool found = false;
        for (int k = 0; k < LANES && !found; ++k)
            found = t.ghost_edge[(int64_t)cu * LANES + k] == ed && t.ghost_prod[(int64_t)cu * LANES + k] == t.edge_prod_cell[ed];
        if (!found) return fail("ghost_prod of stream " + std::to_string(ed));
    }
    return "";
}

// ---------------------------------------------------------------------------------------------- tables <-> file
namespace {
constexpr unsigned long long TABLES_MAGIC = 0x3472546c66485806ull;      // format of flow_tables_save; bump on any change

template <class T>
bool put_vec(FILE *f, const std::vector<T> &v) {
    const unsigned long long n = v.size();
    return fwrite(&n, sizeof(n), 1, f) == 1 && (n == 0 || fwrite(v.data(), sizeof(T), n, f) == n);
}
template <class T>
bool get_vec(FILE *f, std::vector<T> &v) {
    unsigned long long n = 0;
    if (fread(&n, sizeof(n), 1, f) != 1 || n > (1ull << 31)) return false;
    {   // a corrupt count must not turn into a multi-gigabyte allocation: no more elements than the file still holds
        const long here = ftell(f);
        if (here < 0 || fseek(f, 0, SEEK_END) != 0) return false;
        const long end = ftell(f);
        if (end < here || fseek(f, here, SEEK_SET) != 0 || n > (unsigned long long)(end - here) / sizeof(T)) return false;
    }
    v.resize((size_t)n);
    return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}
template <class F>
bool tables_io(FILE *f, FlowTables &t, bool write, F &&vec) {
    int head[13] = {t.n_units, t.n_edges, t.depth, t.n_cells, t.max_imports, t.max_exports, t.skew_ok ? 1 : 0,
                    t.skew_lmax, t.skew_span, t.rsum ? 1 : 0, t.n_folded, t.n_special, t.n_pair_units};
    if (write ? fwrite(head, sizeof(head), 1, f) != 1 : fread(head, sizeof(head), 1, f) != 1) return false;
    if (!write) {
        t.n_units = head[0], t.n_edges = head[1], t.depth = head[2], t.n_cells = head[3], t.max_imports = head[4];
        t.max_exports = head[5], t.skew_ok = head[6] != 0, t.skew_lmax = head[7];
        t.skew_span = head[8], t.rsum = head[9] != 0, t.n_folded = head[10], t.n_special = head[11], t.n_pair_units = head[12];
    }
    return vec(t.cell_of_slot) && vec(t.export_edge) && vec(t.ghost_edge) && vec(t.edge_cons_unit) && vec(t.unit_terms) &&
           vec(t.ent) && vec(t.lag) && vec(t.ghost_lag) && vec(t.unit_p) && vec(t.unit_lmax) && vec(t.unit_glmax) &&
           vec(t.unit_order) && vec(t.ent2) && vec(t.eprev) && vec(t.edge_prod_cell) && vec(t.edge_cons_cell) &&
           vec(t.unit_depth) && vec(t.piece_of_cell) && vec(t.unit_of_cell) && vec(t.height_of_cell) && vec(t.ds) &&
           vec(t.lane_flags) && vec(t.ghost_prod) && vec(t.fold_of_slot);
}
}  // namespace

bool flow_tables_save(const FlowTables &t, const char *path) {
    const std::string tmp = std::string(path) + ".tmp." + std::to_string((long long)getpid());      // ranks may write the same file
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    FlowTables &m = const_cast<FlowTables &>(t);
    bool ok = fwrite(&TABLES_MAGIC, sizeof(TABLES_MAGIC), 1, f) == 1 &&
              tables_io(f, m, true, [&](auto &v) { return put_vec(f, v); });
    ok = fclose(f) == 0 && ok;
    if (ok) ok = rename(tmp.c_str(), path) == 0;
    if (!ok) (void)remove(tmp.c_str());
    return ok;
}

bool flow_tables_load(const char *path, FlowTables &t) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    unsigned long long magic = 0;
    t = FlowTables();
    bool ok = false;
    try {
        ok = fread(&magic, sizeof(magic), 1, f) == 1 && magic == TABLES_MAGIC &&
             tables_io(f, t, false, [&](auto &v) { return get_vec(f, v); }) && fgetc(f) == EOF;
    } catch (...) {      // (std::bad_alloc: the callers fall back to planning)
        ok = false;
    }
    fclose(f);
    if (!ok) t = FlowTables();
    return ok;
}
