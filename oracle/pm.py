"""Penman-Monteith monthly PET (oracle; test infrastructure only).

CPU numpy restatement of xanthos/pet/penman_monteith.py.  Same inputs (the
DataLoader attribute bag, data_load.py:92-135), same arithmetic, written with
broadcasting over ``[land class, cell, month-of-year]`` instead of the
reference's ``np.tile`` copies.  Operation order inside each formula follows
the reference so the two agree to a few ulp.

Reference map:
  year loop / land-cover weighting ...... run_pmpet          :394-477
  per-year preparation .................. SetData.__init__   :17-99
  vegetated surfaces .................... et_veg + calc_*    :102-334
  open water (albedo row 0, emiss 0.98) . et_water           :337-361
  snow / ice (albedo row 6, emiss 0.85) . et_snow            :364-377
"""
import numpy as np

from .months import pm_days_in_month, pm_land_cover_index

LAMBDA1 = 2.46e6     # :76
CP = 1006            # :77
SIGMA = 4.9e-3       # :78
SIGMA2 = 5.67e-8     # :79
GAMMA = 0.67         # :80


def _lc(v):
    """Per-land-class vector -> [L,1,1]."""
    return np.asarray(v, dtype=float)[:, None, None]


def _tab(t):
    """Per-(land class, month) table -> [L,1,12]."""
    return np.asarray(t, dtype=float)[:, None, :]


def _radiation(tair, rsds, rlds, alpha, emiss, dz):
    """Net long-wave and all-wave radiation, J m-2 month-1 (calc_a :155-158, et_water :339-342)."""
    rnl = SIGMA * np.power(tair + 273, 4.0) * emiss * dz - rlds * 86400 * dz
    rn = ((1 - alpha) * rsds) * 86400 * dz - rnl
    return rnl, rn


def pm_year(data, year, st, nlcs, water_idx, snow_idx, land_cover_years):
    """PET for one calendar year: months [st, st+12) of the forcing. Returns [ncell, 12]."""
    ed = st + 12
    tair = data.tair_load[:, st:ed]
    tmin = data.TMIN_load[:, st:ed]
    rhs = data.rhs_load[:, st:ed]
    wind = data.wind_load[:, st:ed]
    rsds = data.rsds_load[:, st:ed]
    rlds = data.rlds_load[:, st:ed]
    tairprev = data.tairprev_load[:, st:ed]

    lc_i = pm_land_cover_index(year, land_cover_years)
    lct = data.lct_load[:, :, lc_i].T[:, :, None]               # [L,C,1]   (:45)
    totpct = np.sum(lct, axis=0)                                # [C,1]
    totpct = np.where(totpct == 0, 0.01, totpct)                # (:47)

    dz = pm_days_in_month(year).astype(float)                   # [12]      (:57-62)
    alpha = _tab(data.alpha)                                    # (:71-73)

    esx = 6.10588 * np.exp(17.32491 * tair / (tair + 238.102))  # (:83)
    vap = esx * (rhs / 100)                                     # (:86, :237)
    sx = 238.1 * 17.325 * esx / np.power(tair + 238.1, 2)       # (:89)
    rsnx = (1 - alpha) * rsds * 86400 * dz                      # (:94)
    wind2 = wind * np.power(2 / 10, 0.11)                       # (:99)

    # ---------------- vegetated surfaces: et_veg (:223-334) ----------------
    p = 101325 * np.power(1 - 0.0065 * data.elev / 288.15, 5.2558)      # [C,1] (:185-188)
    rcorr = p / (101300 * np.power((273.15 + tair) / 293.15, 1.75))     # (:229)
    gcu = 0.00001 * rcorr

    topen, tclose = _lc(data.Tminopen), _lc(data.Tminclose)             # calc_mtmin (:102-114)
    mtmin = np.zeros((nlcs,) + tair.shape)
    mtmin[np.broadcast_to(tmin >= topen, mtmin.shape)] = 1.0
    mtmin[np.broadcast_to(tmin <= tclose, mtmin.shape)] = 0.1
    mid = (tmin < topen) & (tmin > tclose)
    mtmin = np.where(mid, (tmin - tclose) / (topen - tclose), mtmin)

    vopen, vclose = _lc(data.VPDopen), _lc(data.VPDclose)               # calc_vpd (:117-129)
    vpd = esx - vap
    vmid = (vpd > vopen) & (vpd < vclose)
    mvpd = np.broadcast_to(vpd, mtmin.shape)
    mvpd = np.where(vpd <= vopen, 1.0, mvpd)
    mvpd = np.where(vpd >= vclose, 0.1, mvpd)
    mvpd = np.where(vmid, (vclose - vpd) / (vclose - vopen), mvpd)

    gs1 = _lc(data.cL) * mtmin * mvpd * rcorr                           # (:242)

    rh = np.where(rhs > 99.9999, 99.9, rhs)                             # calc_rh (:205-209)

    rblmax, rblmin = _lc(data.RBLmax), _lc(data.RBLmin)                 # calc_rtotc (:132-145)
    rtotc = np.zeros_like(mtmin)
    rtotc = np.where(vpd <= vopen, rblmax, rtotc)
    rtotc = np.where(vpd >= vclose, rblmin, rtotc)
    rtotc = np.where(vmid, rblmax - (rblmax - rblmin) * (vclose - vpd) / (vclose - vopen), rtotc)

    g = 1.6198 * (tair - tairprev)                                      # calc_g (:212-216)
    g[:, 0] = 0

    _, rn = _radiation(tair, rsds, rlds, alpha, _lc(data.emiss), dz)    # calc_a (:148-162)
    a = rn / (86400 * dz)

    lai, laimin, laimax = _tab(data.lai), _tab(data.laimin), _tab(data.laimax)
    fc_denom = np.exp(-0.5 * laimin) - np.exp(-0.5 * laimax)            # (:257-261)
    fc_denom = np.where(fc_denom == 0.0, 1, fc_denom)
    fc = (np.exp(-0.5 * laimin) - np.exp(-0.5 * lai)) / fc_denom
    fc = np.where(fc > 1, 1, fc)

    ac = fc * a
    asoil = (1 - fc) * a - g

    rtot = rtotc * rcorr
    rtot = np.where(rtot > 80, 80, rtot)
    rho = p / ((tair + 273.15) * 287.058)
    rr = rho * CP / (4.0 * SIGMA2 * np.power(tair + 273.15, 3))
    rcx = _lc(data.rc)
    ra = rcx * rr / (rcx + rr)
    ra = np.where(ra > rtot, rtot, ra)

    r100 = rh / 100                                                     # calc_fwet (:165-172)
    fwet = np.where(rh < 70, 0, rh)
    fwet = np.where(rh >= 70, np.power(r100, 8), fwet)
    fwet = np.where(rh >= 80, np.power(r100, 10), fwet)
    fwet = np.where(rh >= 90, np.power(r100, 12), fwet)
    fwet = np.where(rh >= 95, np.power(r100, 16), fwet)

    gsum = gs1 + 1 / rcx + gcu                                          # calc_cc (:192-197)
    cc = np.where(gsum < 0.0001, 10000,
                  np.where(fwet == 1, 0.00001, np.where(lai < 0.0001, 0.00001, 0)))
    with np.errstate(divide='ignore', invalid='ignore'):
        cc = np.where(cc == 0, 1 / rcx * (gs1 + gcu) * lai * (1 - fwet) / gsum, cc)
        rs = np.where(cc == 0, 100000, 1 / cc)                          # (:285-291)
    rslimit = _lc(data.rslimit)
    rs = np.where(rs > rslimit, rslimit, rs)

    lai_fwet = np.where(lai * fwet == 0, 1, lai * fwet)                 # (:296-301)
    rhc = np.where(lai > 0.00001, rcx / lai_fwet, rslimit)
    rhc = np.where(rhc > rslimit, rslimit, rhc)
    rvc = rhc
    rhrc = rhc * rr / (rhc + rr)
    rhrc = np.where(rhrc > rtot, rtot, rhrc)

    apres = dz * 86400 * (sx * ac + rho * CP * vpd * fc / rhrc) * fwet / (
        (sx + p * 0.01 * CP * rvc / (LAMBDA1 * 0.622 * rhrc)) * LAMBDA1)    # (:306-307)
    ewet_c = np.where(rh >= 70, apres, 0.0)

    rasoil = rtot * rr / (rtot + rr)
    soil_num = 86400 * dz * (sx * asoil + rho * CP * (1 - fc) * vpd / rasoil)
    soil_den = (sx + GAMMA * rtot / rasoil) * LAMBDA1
    ewet_soil = soil_num * fwet / soil_den                              # (:314-315)
    esoilpot = soil_num * (1 - fwet) / soil_den                         # (:316-317)
    esoil = ewet_soil + esoilpot * np.power(r100, vpd / _lc(data.beta))    # (:323)

    trans = dz * 86400 * (sx * ac + rho * CP * vpd * fc / ra) * (1 - fwet) / (
        (sx + GAMMA * (1 + rs / ra)) * LAMBDA1)                         # (:326-327)
    trans = np.where(fc == 0, 0, trans)

    eet = trans + ewet_c + esoil
    arr = np.where(eet < 0.0, 0.0, eet)                                 # [L,C,12]

    # ---------------- open water: et_water (:337-361), albedo row 0 ----------------
    a0 = np.asarray(data.alpha, dtype=float)[0]
    rnl_w, rn_w = _radiation(tair, rsds, rlds, a0, 0.98, dz)
    rn_w = np.where(rn_w < 0, 0.0, rn_w)
    mth = np.arange(12)
    qt = 0.5 * rsnx[0] - np.where(mth <= 5, 0.8, 1.3) * rnl_w
    ax = (rn_w - qt) / (86400 * dz)
    ax = np.where(ax < 0, 0, ax)
    ewetx = rn_w / (86400 * dz) * dz * 0.6 / 2845
    ewety = dz * 86400 * (sx * ax + GAMMA * 6.43 * (0.5 + 0.54 * wind2) * (esx - vap)) / (
        (sx + GAMMA) * LAMBDA1)
    wat = np.where(tair < -1, ewetx, ewety)
    wat = np.where(wat < 0.0, 0.0, wat)
    arr[water_idx] = wat                                                # (:460)

    # ---------------- snow / ice: et_snow (:364-377), albedo row 6 ----------------
    a6 = np.asarray(data.alpha, dtype=float)[6]
    _, rn_s = _radiation(tair, rsds, rlds, a6, 0.85, dz)
    rn_s = np.where(rn_s < 0, 0.0, rn_s)
    snw = rn_s / (86400 * dz) * dz * 0.6 / 2845
    snw = np.where(snw < 0.0, 0.0, snw)
    arr[snow_idx] = snw                                                 # (:464)

    arr = arr * lct                                                     # (:467)
    return np.sum(arr, axis=0) / totpct                                 # (:470)


def run_pmpet(data, ncells, nlcs, start_yr, end_yr, water_idx, snow_idx, land_cover_years):
    """Same signature and result as penman_monteith.run_pmpet (:394-477)."""
    nyears = end_yr - start_yr + 1
    out = np.zeros((ncells, 12 * nyears))
    for k, year in enumerate(range(start_yr, end_yr + 1)):
        out[:, 12 * k:12 * k + 12] = pm_year(data, year, 12 * k, nlcs, water_idx, snow_idx,
                                             land_cover_years)
    return out
