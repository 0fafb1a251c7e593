// Host-side river-network topology for MRTM (integer work; runs without a device).
//
// Replaces xanthos/routing/mrtm.py: downstream + make_flowdirgrid (:85-120, :233-258), upstream (:123-191) and
// upstream_genmatrix (:194-230).  The reference rebuilds these with numpy on every calculate_routing call
// (components.py:268-270; ~0.2 s); here they are three linear passes over the cells.
#include <algorithm>
#include <vector>

#include "xh_common.h"

namespace {

// bit groups of the D8 code (mrtm.py:236-240)
constexpr int D8_RIGHT = 1 | 2 | 128, D8_LEFT = 8 | 16 | 32, D8_UP = 32 | 64 | 128, D8_DOWN = 2 | 4 | 8;

int build_grid(int64_t ncell, int nrow, int ncol, const int64_t *id, const int32_t *ilon, const int32_t *ilat,
               std::vector<int64_t> &grid) {
    grid.assign((size_t)nrow * ncol, 0);
    for (int64_t i = 0; i < ncell; ++i) {
        const int r = ilat[i] - 1, c = ilon[i] - 1;
        if (r < 0 || r >= nrow || c < 0 || c >= ncol) return XH_ERR_ARG;
        grid[(size_t)r * ncol + c] = id[i];
    }
    return XH_OK;
}

}  // namespace

extern "C" int xh_mrtm_downstream(int64_t ncell, int32_t nrow, int32_t ncol, const int64_t *h_id,
                                  const int32_t *h_ilon, const int32_t *h_ilat, const double *h_flowdir,
                                  int64_t *h_dsid) {
    if (!h_id || !h_ilon || !h_ilat || !h_flowdir || !h_dsid || ncell < 0 || nrow <= 0 || ncol <= 0)
        return xh_fail(nullptr, XH_ERR_ARG, "xh_mrtm_downstream: bad argument");
    std::vector<int64_t> grid;
    if (build_grid(ncell, nrow, ncol, h_id, h_ilon, h_ilat, grid))
        return xh_fail(nullptr, XH_ERR_ARG, "xh_mrtm_downstream: cell outside the %d x %d grid", nrow, ncol);
    for (int64_t i = 0; i < ncell; ++i) {
        const int r0 = h_ilat[i] - 1, c0 = h_ilon[i] - 1;
        const int code = h_flowdir[i] == -9999.0 ? 0 : (int)h_flowdir[i];            // :243-245
        int dr = 0, dc = 0;
        if (code & D8_DOWN) dr = -1;
        if (code & D8_UP) dr = 1;                                                   // 'up' wins over 'down' (:254-255)
        if (code & D8_RIGHT) dc = 1;
        if (code & D8_LEFT) dc = -1;                                                // 'left' wins over 'right' (:256-257)
        int r = r0 + dr, c = c0 + dc;
        if (c < 0 || c > ncol - 1) c = ((c + 1) % ncol + ncol) % ncol;              // :100-102 (python mod)
        if (r < 0 || r > nrow - 1) {                                                // :104-106
            r = r0;
            c = c0;
        }
        const int64_t t = grid[(size_t)r * ncol + c];
        h_dsid[i] = (t == 0 || t == h_id[i]) ? -1 : t;                              // :116-118
    }
    return XH_OK;
}

extern "C" int xh_mrtm_upstream(int64_t ncell, int32_t nrow, int32_t ncol, const int64_t *h_id, const int32_t *h_ilon,
                                const int32_t *h_ilat, const int64_t *h_dsid, int64_t *h_upid) {
    if (!h_id || !h_ilon || !h_ilat || !h_dsid || !h_upid || ncell < 0 || nrow <= 0 || ncol <= 0)
        return xh_fail(nullptr, XH_ERR_ARG, "xh_mrtm_upstream: bad argument");
    std::vector<int64_t> grid;
    if (build_grid(ncell, nrow, ncol, h_id, h_ilon, h_ilat, grid))
        return xh_fail(nullptr, XH_ERR_ARG, "xh_mrtm_upstream: cell outside the %d x %d grid", nrow, ncol);
    static const int roff[8] = {-1, -1, -1, 0, 0, 1, 1, 1}, coff[8] = {-1, 0, 1, -1, 1, -1, 0, 1};   // :142-143
    for (int64_t i = 0; i < ncell; ++i) {
        int64_t nb[8];
        bool in[8];
        for (int k = 0; k < 8; ++k) {
            const int r = h_ilat[i] - 1 + roff[k], c = h_ilon[i] - 1 + coff[k];
            nb[k] = (r >= 0 && c >= 0 && r <= nrow - 1 && c <= ncol - 1) ? grid[(size_t)r * ncol + c] : 0;   // no wrap (:150)
            in[k] = false;
            if (nb[k] != 0) {
                if (nb[k] < 1 || nb[k] > ncell)
                    return xh_fail(nullptr, XH_ERR_ARG, "xh_mrtm_upstream: ids must be 1..ncell");
                in[k] = h_dsid[nb[k] - 1] == h_id[i];                               // :164-166
            }
        }
        int64_t *row = h_upid + i * 9;
        int w = 0;
        for (int k = 0; k < 8; ++k)
            if (in[k]) row[w++] = nb[k];                                            // inflowing first, stable (:169-184)
        const int cnt = w;
        for (int k = 0; k < 8; ++k)
            if (!in[k]) row[w++] = nb[k];
        row[8] = cnt;
    }
    return XH_OK;
}

extern "C" int xh_mrtm_um_csr(int64_t ncell, const int64_t *h_upid, int64_t *h_indptr, int32_t *h_indices,
                              int8_t *h_sign) {
    if (!h_upid || !h_indptr || !h_indices || !h_sign || ncell < 0)
        return xh_fail(nullptr, XH_ERR_ARG, "xh_mrtm_um_csr: bad argument");
    int64_t pos = 0;
    h_indptr[0] = 0;
    for (int64_t i = 0; i < ncell; ++i) {
        const int64_t *row = h_upid + i * 9;
        const int cnt = (int)row[8];
        int64_t cols[9];
        int n = 0;
        for (int k = 0; k < cnt; ++k) cols[n++] = row[k] - 1;
        cols[n++] = i;                                                              // the -I diagonal (:228)
        std::sort(cols, cols + n);
        for (int k = 0; k < n; ++k) {
            if (cols[k] < 0 || cols[k] >= ncell)
                return xh_fail(nullptr, XH_ERR_ARG, "xh_mrtm_um_csr: upstream id out of range");
            h_indices[pos] = (int32_t)cols[k];
            h_sign[pos] = cols[k] == i ? -1 : 1;
            ++pos;
        }
        h_indptr[i + 1] = pos;
    }
    return XH_OK;
}
