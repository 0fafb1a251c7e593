// Host-only fuzzer of the routing planner (xanthos_amd/csrc/xh_flow_plan.cpp), built with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all
// Random river networks -- D8-like forests on grids, chains, stars, single cells, 10^5 cells -- and inputs that are NOT
// trees (cycles, cells with two downstream rows, rows without a diagonal), with and without a set of cells that can
// fire (the reassociated planner's single-sum plans), under several planner options.  Every plan must pass
// flow_tables_check: every cell in exactly one slot, streams strictly down the pipeline, <= 16 imports / outlets per unit,
// even lane lags consistent with the "two iterations earlier" rule, every row -- expanded through its chains -- equal to the
// CSR row in stored order (mrtm.py:50-51); the reassociated plans flow_tables_check_rsum.
//   usage: plan_fuzz [cases] [seed]
#include <algorithm>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <chrono>
#include <string>
#include <vector>

#include "xh_flow_plan.h"

struct Graph {
    int n = 0;
    std::vector<int64_t> indptr;
    std::vector<int32_t> indices;
    std::vector<int8_t> sign;
    std::vector<int> comp;
    int ncomp = 0;
    std::vector<char> tree_cell;      // 1 = the cell's component is a plain tree (the planner must route it)
};

static void finish(Graph &g, const std::vector<std::vector<int>> &up, const std::vector<char> &drop_diag) {
    const int n = g.n;
    g.indptr.assign(n + 1, 0);
    g.indices.clear();
    g.sign.clear();
    for (int c = 0; c < n; ++c) {
        std::vector<int> row(up[c]);
        std::sort(row.begin(), row.end());
        bool diag_done = drop_diag[c] != 0;
        for (int s : row) {
            if (!diag_done && s > c) {
                g.indices.push_back(c);
                g.sign.push_back(-1);
                diag_done = true;
            }
            g.indices.push_back(s);
            g.sign.push_back(1);
        }
        if (!diag_done) {
            g.indices.push_back(c);
            g.sign.push_back(-1);
        }
        g.indptr[c + 1] = (int64_t)g.indices.size();
    }
    std::vector<int> parent(n);
    std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](int x) {
        while (parent[x] != x) x = parent[x] = parent[parent[x]];
        return x;
    };
    for (int c = 0; c < n; ++c)
        for (int64_t j = g.indptr[c]; j < g.indptr[c + 1]; ++j) {
            const int a = find(c), b = find(g.indices[j]);
            if (a != b) parent[std::max(a, b)] = std::min(a, b);
        }
    g.comp.assign(n, 0);
    std::vector<int> id(n, -1);
    g.ncomp = 0;
    for (int c = 0; c < n; ++c) {
        const int r = find(c);
        if (id[r] < 0) id[r] = g.ncomp++;
        g.comp[c] = id[r];
    }
}

// forest on a w x h grid: every cell drains to a random lower neighbour (D8) or nowhere; `bad` plants defects
static Graph make_grid(std::mt19937_64 &rng, int w, int h, double p_outlet, int bad, double slope) {
    Graph g;
    g.n = w * h;
    std::vector<double> elev(g.n);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    // slope 0: pure noise (many tiny networks); larger: the terrain falls towards a corner and the networks get long
    const double ax = U(rng) - 0.3, ay = U(rng) - 0.3;
    for (int c = 0; c < g.n; ++c) elev[c] = U(rng) + slope * (ax * (c % w) + ay * (c / w));
    std::vector<int> ds(g.n, -1);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int c = y * w + x;
            if (U(rng) < p_outlet) continue;
            int cand[8], nc = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (!dx && !dy) continue;
                    const int xx = x + dx, yy = y + dy;
                    if (xx < 0 || yy < 0 || xx >= w || yy >= h) continue;
                    if (elev[yy * w + xx] < elev[c]) cand[nc++] = yy * w + xx;
                }
            if (nc) ds[c] = cand[rng() % nc];
        }
    std::vector<std::vector<int>> up(g.n);
    for (int c = 0; c < g.n; ++c)
        if (ds[c] >= 0) up[ds[c]].push_back(c);
    std::vector<char> drop(g.n, 0), defect_cell(g.n, 0);
    for (int k = 0; k < bad && g.n >= 4; ++k) {
        const int c = (int)(rng() % g.n);
        switch (rng() % 3) {
            case 0: {       // a second downstream row for c
                const int d = (int)(rng() % g.n);
                if (d != c && std::find(up[d].begin(), up[d].end(), c) == up[d].end() && up[d].size() < 8) up[d].push_back(c);
                defect_cell[c] = 1;
                defect_cell[d] = 1;
                break;
            }
            case 1: {       // a two-cycle
                const int d = (int)(rng() % g.n);
                if (d != c && up[c].size() < 8 && up[d].size() < 8) {
                    if (std::find(up[d].begin(), up[d].end(), c) == up[d].end()) up[d].push_back(c);
                    if (std::find(up[c].begin(), up[c].end(), d) == up[c].end()) up[c].push_back(d);
                }
                defect_cell[c] = 1;
                defect_cell[d] = 1;
                break;
            }
            default:        // a row without its diagonal
                drop[c] = 1;
                defect_cell[c] = 1;
        }
    }
    finish(g, up, drop);
    // which components are still trees: recompute honestly (each cell in <= 1 row as +1, one diagonal, no cycle)
    std::vector<char> comp_ok(g.ncomp, 1);
    std::vector<int> d2(g.n, -1);
    for (int r = 0; r < g.n; ++r) {
        int nd = 0;
        for (int64_t j = g.indptr[r]; j < g.indptr[r + 1]; ++j) {
            const int c = g.indices[j];
            if (g.sign[j] < 0) nd += (c == r);
            else {
                if (d2[c] >= 0 || c == r) comp_ok[g.comp[r]] = 0;
                d2[c] = r;
            }
        }
        if (nd != 1 || g.indptr[r + 1] - g.indptr[r] > 9) comp_ok[g.comp[r]] = 0;
    }
    for (int s = 0; s < g.n; ++s) {      // cycles
        int v = s, steps = 0;
        while (v >= 0 && steps <= g.n) {
            v = d2[v];
            ++steps;
        }
        if (steps > g.n) comp_ok[g.comp[s]] = 0;
    }
    g.tree_cell.assign(g.n, 0);
    for (int c = 0; c < g.n; ++c) g.tree_cell[c] = comp_ok[g.comp[c]];
    (void)defect_cell;
    return g;
}

// the tables of a plan as one number (PLAN_HASH=1: to compare two builds of the planner)
static unsigned long long tables_hash(const FlowTables &t) {
    unsigned long long hsh = 1469598103934665603ull;
    auto mix = [&](const void *q, size_t bytes) {
        const unsigned char *b = static_cast<const unsigned char *>(q);
        for (size_t i = 0; i < bytes; ++i) hsh = (hsh ^ b[i]) * 1099511628211ull;
    };
    auto mv = [&](const auto &v) { if (!v.empty()) mix(v.data(), v.size() * sizeof(v[0])); };
    int head[8] = {t.n_units, t.n_edges, t.depth, t.n_cells, t.max_imports, t.max_exports, t.skew_lmax, t.skew_span};
    mix(head, sizeof(head));
    mv(t.cell_of_slot), mv(t.export_edge), mv(t.ghost_edge), mv(t.edge_cons_unit), mv(t.unit_terms), mv(t.ent), mv(t.lag), mv(t.ghost_lag);
    mv(t.unit_p), mv(t.unit_lmax), mv(t.unit_glmax), mv(t.unit_order), mv(t.ent2), mv(t.eprev), mv(t.edge_prod_cell), mv(t.edge_cons_cell);
    mv(t.unit_depth), mv(t.piece_of_cell), mv(t.unit_of_cell), mv(t.height_of_cell), mv(t.ds), mv(t.ghost_prod), mv(t.lane_flags);
    return hsh;
}

static int run_case(std::mt19937_64 &rng, int idx, bool verbose) {
    const int kind = idx % 11;
    int w, h, bad = 0;
    double p_out = 0.02, slope = (rng() % 3 == 0) ? 0.0 : 0.2 * (double)(1 + rng() % 10);
    switch (kind) {
        case 0: w = 1; h = 1; break;                                   // a single cell
        case 1: w = 1; h = 2 + (int)(rng() % 400); p_out = 0.0; slope = 10.0; break;  // chains (1 x h: the only lower neighbour is along the line)
        case 2: w = 3; h = 3; p_out = 0.0; break;                       // a star at most
        case 3: w = 200 + (int)(rng() % 120); h = 316; p_out = 0.01; break;      // ~10^5 cells
        case 4: w = 40; h = 30; bad = 1 + (int)(rng() % 6); break;      // defects
        case 5: w = 64; h = 64; p_out = 0.3; break;                     // many tiny networks
        default: w = 8 + (int)(rng() % 90); h = 8 + (int)(rng() % 90); p_out = 0.002 + 0.05 * (double)(rng() % 100) / 100.0;
    }
    Graph g = make_grid(rng, w, h, p_out, bad, slope);
    FlowPlanOptions opt;
    opt.simds = (rng() % 3 == 0) ? 0 : 1024;
    static const int caps[] = {0, 0, 64, 36, 20, 12, 5};
    opt.piece_cap = caps[rng() % 7];
    opt.chain = rng() % 4 != 0;
    opt.cut_rule = rng() % 4 != 0;
    opt.tlimit = 3 + (int)(rng() % 7);
    for (int k = 0; k < 4; ++k) (void)rng();      // (options of the typed partitions, gone in round 6: the cases stay what they were)
    if (rng() % 2 == 0) (void)rng();
    opt.lane_trials = idx % 3 == 1 ? 60 : 0;                         // lanes re-assigned against bank conflicts in a third
    std::vector<unsigned char> capable;
    if (rng() % 3 != 0) {
        capable.assign(g.n, 0);
        const unsigned pct = (unsigned)(rng() % 4 == 0 ? 30 : rng() % 4);      // 0-3 %, sometimes 30 %
        for (int c = 0; c < g.n; ++c) capable[c] = (rng() % 100) < pct;
        opt.capable = capable.data();
    }
    std::vector<char> handled;
    FlowTables t;
    std::string err;
    if (flow_tables_build(g.n, g.indptr.data(), g.indices.data(), g.sign.data(), g.comp.data(), g.ncomp, opt, handled, t, err) != 0) {
        fprintf(stderr, "case %d (kind %d, %d cells): build failed: %s\n", idx, kind, g.n, err.c_str());
        return 1;
    }
    if (getenv("PLAN_HASH")) printf("hash %d %016llx\n", idx, tables_hash(t));      // (to compare two builds of the planner)
    for (int c = 0; c < g.n; ++c)
        if ((handled[c] != 0) != (g.tree_cell[c] != 0)) {
            fprintf(stderr, "case %d (kind %d, %d cells): cell %d handled %d, tree %d\n", idx, kind, g.n, c, (int)handled[c],
                    (int)g.tree_cell[c]);
            return 1;
        }
    const std::string bad_msg = flow_tables_check(g.n, g.indptr.data(), g.indices.data(), g.sign.data(), handled, t);
    if (!bad_msg.empty()) {
        fprintf(stderr, "case %d (kind %d, %d cells, cap %d, chain %d, cut %d): %s\n", idx, kind, g.n, opt.piece_cap,
                (int)opt.chain, (int)opt.cut_rule, bad_msg.c_str());
        return 1;
    }
    // The per-box cache (xh_route_plan_prepare, flow_plan_build) keeps tables in a file: what comes back must be what went
    // in -- and pass the same checker --, and a truncated file must be refused.
    if (t.n_units > 0 && (idx % 4) == 0) {
        char path[128];
        snprintf(path, sizeof(path), "/tmp/plan_fuzz_%d_%d.tables", (int)getpid(), idx);
        FlowTables u;
        bool ok = flow_tables_save(t, path) && flow_tables_load(path, u);
        ok = ok && u.n_units == t.n_units && u.n_edges == t.n_edges && u.depth == t.depth &&
             u.skew_ok == t.skew_ok && u.skew_lmax == t.skew_lmax && u.skew_span == t.skew_span && u.cell_of_slot == t.cell_of_slot &&
             u.ent2 == t.ent2 && u.eprev == t.eprev && u.lag == t.lag && u.ghost_lag == t.ghost_lag && u.unit_p == t.unit_p &&
             u.unit_order == t.unit_order && u.export_edge == t.export_edge && u.ghost_edge == t.ghost_edge &&
             u.edge_cons_unit == t.edge_cons_unit && u.lane_flags == t.lane_flags && u.ghost_prod == t.ghost_prod &&
             u.ent == t.ent;
        ok = ok && flow_tables_check(g.n, g.indptr.data(), g.indices.data(), g.sign.data(), handled, u).empty();
        if (ok) {      // cut the file short: the loader has to notice
            FILE *f = fopen(path, "rb+");
            if (f) {
                fseek(f, 0, SEEK_END);
                const long len = ftell(f);
                fclose(f);
                if (len > 16 && truncate(path, len - 7) == 0) ok = !flow_tables_load(path, u);
            }
        }
        remove(path);
        if (!ok) {
            fprintf(stderr, "case %d (kind %d, %d cells): tables did not survive the round trip through a file\n", idx, kind, g.n);
            return 1;
        }
    }
    if (verbose)
        printf("case %d kind %d: %d cells, %d units, %d streams, depth %d, max lag %d\n", idx, kind, g.n, t.n_units, t.n_edges, t.depth,
               t.skew_lmax);
    // The reassociated form of the same graph (xh_flow_rsum.cpp): the same cells routed, its own invariants, the file round trip
    {
        std::vector<char> handled_r;
        FlowTables r;
        // leaves that may be carried by their downstream cell's lane (half of the cases): cells without upstream neighbours that
        // cannot fire
        std::vector<unsigned char> foldable;
        FlowPlanOptions opt_r = opt;
        if (idx % 2 == 0) {
            foldable.assign(g.n, 0);
            for (int c = 0; c < g.n; ++c)
                foldable[c] = (g.indptr[c + 1] - g.indptr[c] == 1) && (capable.empty() || !capable[c]) && (rng() % 8 != 0);
            opt_r.foldable = foldable.data();
        }
        // single-sum plans: the cells that can fire as drawn above (two thirds of the cases), or drawn here at 2 / 10 / 40 %
        std::vector<unsigned char> capable_r;
        if (opt_r.capable == nullptr && idx % 3 != 2) {
            const unsigned dens = (idx % 3 == 0) ? 2u : ((idx / 3) % 2 ? 10u : 40u);
            capable_r.assign(g.n, 0);
            for (int c = 0; c < g.n; ++c) capable_r[c] = (rng() % 100u < dens) && !(opt_r.foldable && foldable[c]);
            opt_r.capable = capable_r.data();
        }
        if (flow_tables_build_rsum(g.n, g.indptr.data(), g.indices.data(), g.sign.data(), g.comp.data(), g.ncomp, opt_r, handled_r, r, err) != 0) {
            fprintf(stderr, "case %d (kind %d, %d cells): reassociated build failed: %s\n", idx, kind, g.n, err.c_str());
            return 1;
        }
        if (handled_r != handled) {
            fprintf(stderr, "case %d (kind %d, %d cells): the reassociated plan routes other cells than the bit-exact one\n", idx, kind, g.n);
            return 1;
        }
        std::string bad_r = flow_tables_check_rsum(g.n, g.indptr.data(), g.indices.data(), g.sign.data(), handled_r, r);
        if (bad_r.empty() && r.n_units > 0 && (r.skew_lmax > 192 || r.max_imports > 16 || r.max_exports > 16)) bad_r = "limits";
        if (bad_r.empty() && r.n_units > 0 && (idx % 4) == 1) {
            char path[128];
            snprintf(path, sizeof(path), "/tmp/plan_fuzz_%d_%d.rtables", (int)getpid(), idx);
            FlowTables u;
            if (!(flow_tables_save(r, path) && flow_tables_load(path, u) && u.rsum && u.ent2 == r.ent2 && u.eprev == r.eprev &&
                  u.lag == r.lag && u.unit_p == r.unit_p && u.fold_of_slot == r.fold_of_slot && u.n_folded == r.n_folded && u.n_special == r.n_special && u.lane_flags == r.lane_flags &&
                  flow_tables_check_rsum(g.n, g.indptr.data(), g.indices.data(), g.sign.data(), handled_r, u).empty()))
                bad_r = "tables did not survive the round trip through a file";
            remove(path);
        }
        if (!bad_r.empty()) {
            fprintf(stderr, "case %d (kind %d, %d cells, cap %d): reassociated plan: %s\n", idx, kind, g.n, opt.piece_cap, bad_r.c_str());
            return 1;
        }
        if (verbose)
            printf("        reassociated: %d units, %d streams, depth %d, max lag %d, %d leaves folded, %d special cells\n", r.n_units, r.n_edges, r.depth, r.skew_lmax, r.n_folded, r.n_special);
    }
    return 0;
}

// plan_fuzz --file topo.bin [ignored] [reassociated 0|1] : the planner on a topology written by tools/dump_topology.py
// (int32 n, int64 nnz, indptr[n+1] int64, indices[nnz] int32, sign[nnz] int8, capable[n] uint8), with statistics
static int run_file(const char *path, bool rsum) {
    FILE *f = fopen(path, "rb");
    if (!f) return 2;
    int32_t n = 0;
    int64_t nnz = 0;
    if (fread(&n, 4, 1, f) != 1 || fread(&nnz, 8, 1, f) != 1) return 2;
    Graph g;
    g.n = n;
    g.indptr.resize(n + 1);
    g.indices.resize(nnz);
    g.sign.resize(nnz);
    std::vector<unsigned char> capable(n);
    if (fread(g.indptr.data(), 8, n + 1, f) != (size_t)n + 1 || fread(g.indices.data(), 4, nnz, f) != (size_t)nnz ||
        fread(g.sign.data(), 1, nnz, f) != (size_t)nnz || fread(capable.data(), 1, n, f) != (size_t)n)
        return 2;
    fclose(f);
    std::vector<int> parent(n);
    std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](int x) {
        while (parent[x] != x) x = parent[x] = parent[parent[x]];
        return x;
    };
    for (int c = 0; c < n; ++c)
        for (int64_t j = g.indptr[c]; j < g.indptr[c + 1]; ++j) {
            const int a = find(c), b = find(g.indices[j]);
            if (a != b) parent[std::max(a, b)] = std::min(a, b);
        }
    g.comp.assign(n, 0);
    std::vector<int> id(n, -1);
    for (int c = 0; c < n; ++c) {
        const int r = find(c);
        if (id[r] < 0) id[r] = g.ncomp++;
        g.comp[c] = id[r];
    }
    FlowPlanOptions opt;
    opt.simds = 1024;
    opt.debug = true;
    if (getenv("TLIMIT")) opt.tlimit = atoi(getenv("TLIMIT"));
    if (getenv("PIECE_CAP")) opt.piece_cap = atoi(getenv("PIECE_CAP"));
    if (getenv("LANE_TRIALS")) opt.lane_trials = atoi(getenv("LANE_TRIALS"));
    int ncap = 0;
    for (unsigned char c : capable) ncap += c;
    printf("%d cells, %lld entries, %d cells can fire\n", n, (long long)nnz, ncap);
    std::vector<char> handled;
    FlowTables t;
    std::string err;
    if (getenv("SIMDS")) opt.simds = atoi(getenv("SIMDS"));
    std::vector<unsigned char> foldable;
    if (rsum && getenv("FOLD")) {
        foldable.assign(n, 0);
        for (int c = 0; c < n; ++c) foldable[c] = (g.indptr[c + 1] - g.indptr[c] == 1) && !capable[c];
        opt.foldable = foldable.data();
    }
    if (rsum) opt.capable = getenv("SINGLE") ? capable.data() : nullptr;
    if (getenv("HALO")) opt.halo = atoi(getenv("HALO"));
    if (getenv("PAIR_IMPORTS")) opt.pair_imports = atoi(getenv("PAIR_IMPORTS"));
    const auto t0 = std::chrono::steady_clock::now();
    if (rsum ? flow_tables_build_rsum(n, g.indptr.data(), g.indices.data(), g.sign.data(), g.comp.data(), g.ncomp, opt, handled, t, err)
             : flow_tables_build(n, g.indptr.data(), g.indices.data(), g.sign.data(), g.comp.data(), g.ncomp, opt, handled, t, err)) {
        printf("build failed: %s\n", err.c_str());
        return 1;
    }
    printf("flow_tables_build%s: %.1f ms\n", rsum ? "_rsum" : "", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    const std::string bad = rsum ? flow_tables_check_rsum(n, g.indptr.data(), g.indices.data(), g.sign.data(), handled, t)
                                 : flow_tables_check(n, g.indptr.data(), g.indices.data(), g.sign.data(), handled, t);
    printf("check: %s\n", bad.empty() ? "ok" : bad.c_str());
    if (getenv("PLAN_HASH")) printf("hash file %016llx\n", tables_hash(t));
    if (const char *dc = getenv("DUMP_CELLS")) {      // DUMP_CELLS=a,b,c: where the planner put these cells
        std::vector<int> slot_of(n, -1);
        for (size_t sl = 0; sl < t.cell_of_slot.size(); ++sl)
            if (t.cell_of_slot[sl] >= 0) slot_of[t.cell_of_slot[sl]] = (int)sl;
        const int64_t ts = (int64_t)t.n_units * 64;
        auto ent_name = [&](int u, unsigned off) {
            char buf[96];
            const int e = (int)(off / 16u);
            if (e == 128) snprintf(buf, sizeof(buf), "zero");
            else if (e >= 64) snprintf(buf, sizeof(buf), "import %d (stream %d from cell %d)", e - 64, t.ghost_edge[(int64_t)u * 64 + e - 64], t.ghost_prod[(int64_t)u * 64 + e - 64]);
            else snprintf(buf, sizeof(buf), "lane %d (cell %d)", e, t.cell_of_slot[(int64_t)u * 64 + e]);
            return std::string(buf);
        };
        for (const char *q = dc; *q;) {
            const int c = atoi(q);
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
            if (c < 0 || c >= n || slot_of[c] < 0) { printf("cell %d: no slot\n", c); continue; }
            const int sl = slot_of[c], u = sl / 64;
            int ncell = 0, nimp = 0, nexp = 0;
            for (int k = 0; k < 64; ++k) {
                ncell += t.cell_of_slot[(int64_t)u * 64 + k] >= 0;
                nimp += t.ghost_edge[(int64_t)u * 64 + k] >= 0;
                nexp += t.export_edge[(int64_t)u * 64 + k] >= 0;
            }
            printf("cell %d: unit %d lane %d shape 0x%x (cells %d imports %d outlets %d depth %d lmax %d) lag %d flag %d A = %s, R = %s, export %d\n", c, u, sl % 64,
                   t.unit_p[u], ncell, nimp, nexp, t.unit_depth[u], t.unit_lmax[u], t.lag[sl], (int)t.lane_flags[sl], ent_name(u, t.ent2[(size_t)sl]).c_str(),
                   ent_name(u, t.eprev[sl]).c_str(), t.export_edge[sl]);
            (void)ts;
        }
    }
    return bad.empty() ? 0 : 1;
}

int main(int argc, char **argv) {
    if (argc > 2 && std::string(argv[1]) == "--file") return run_file(argv[2], argc > 4 && atoi(argv[4]) != 0);
    const int cases = argc > 1 ? atoi(argv[1]) : 200;
    const unsigned long long seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 20240807ull;
    std::mt19937_64 rng(seed);
    int failed = 0;
    for (int i = 0; i < cases; ++i) failed += run_case(rng, i, argc > 3);
    printf("plan_fuzz: %d cases, %d failed\n", cases, failed);
    return failed ? 1 : 0;
}
