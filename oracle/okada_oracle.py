"""
TEST INFRASTRUCTURE ONLY -- analytic half-space geodetic forward models (numpy).

BASELINE configs 1/2 (Mogi / rectangular-source geodetic composites) run, in the reference,
through pyrocko's GF-store engine (heart.geo_synthetics, beat/heart.py:4158-4239): that
arithmetic is NOT in the reference tree and pyrocko is not installed -> parity with BEAT is
UNPINNED for geometry mode.  The build's counterpart is the analytic homogeneous
half-space solution; this file restates the published formulas and is pinned to the
published check values:

  Okada, Y. (1985), Surface deformation due to shear and tensile faults in a half-space,
  BSSA 75(4), 1135-1154: eqs (25)-(30); Table 2 "checklist for numerical calculations", case 2
  (x=2, y=3, d=4, dip=70 deg, L=3, W=2, lambda=mu), finite source:
     strike-slip  ux=-8.689e-3  uy=-4.298e-3  uz=-2.747e-3
     dip-slip     ux=-4.682e-3  uy=-3.527e-2  uz=-3.564e-2
     tensile      ux=-2.660e-4  uy=+1.056e-2  uz=+3.214e-3
  Mogi, K. (1958): u = (1-nu)/pi * dV * (x, y, d) / R^3.

Conventions of the rectangular source (pyrocko/BEAT RectangularSource, anchor at the centre of
the TOP edge): east_shift, north_shift, depth [km] of that point; strike, dip, rake [deg];
length along strike, width down-dip [km]; slip [m]; opening_fraction f in [-1,1]:
shear = slip*(1-|f|) split by rake into strike/dip components, opening = slip*f.
"""
import numpy as np


def _chinnery(f, x, p, L, W, q, dip, nu):
    return (f(x, p, q, dip, nu) - f(x, p - W, q, dip, nu)
            - f(x - L, p, q, dip, nu) + f(x - L, p - W, q, dip, nu))


def _common(xi, eta, q, dip):
    sd, cd = np.sin(dip), np.cos(dip)
    R = np.sqrt(xi ** 2 + eta ** 2 + q ** 2)
    yt = eta * cd + q * sd
    dt = eta * sd - q * cd
    return sd, cd, R, yt, dt


def _I(xi, eta, q, dip, nu, R, yt, dt):
    """Okada (1985) eqs (28)-(29): I1..I5 with mu/(lambda+mu) = 1-2nu"""
    sd, cd = np.sin(dip), np.cos(dip)
    a = 1.0 - 2.0 * nu
    X = np.sqrt(xi ** 2 + q ** 2)
    if np.abs(cd) > 1e-12:
        with np.errstate(divide="ignore", invalid="ignore"):
            I5 = a * 2.0 / cd * np.arctan((eta * (X + q * cd) + X * (R + X) * sd)
                                          / (xi * (R + X) * cd))
        I5 = np.where(np.abs(xi) < 1e-12, 0.0, I5)
        I4 = a / cd * (np.log(R + dt) - sd * np.log(R + eta))
        I3 = a * (yt / (cd * (R + dt)) - np.log(R + eta)) + sd / cd * I4
        I1 = a * (-xi / (cd * (R + dt))) - sd / cd * I5
    else:
        I5 = -a * xi * sd / (R + dt)
        I4 = -a * q / (R + dt)
        I3 = a / 2.0 * (eta / (R + dt) + yt * q / (R + dt) ** 2 - np.log(R + eta))
        I1 = -a / 2.0 * xi * q / (R + dt) ** 2
    I2 = a * (-np.log(R + eta)) - I3
    return I1, I2, I3, I4, I5


def _atan_term(xi, eta, q, R):
    with np.errstate(divide="ignore", invalid="ignore"):
        v = np.arctan(xi * eta / (q * R))
    return np.where(np.abs(q) < 1e-12, 0.0, v)


def _ss(xi, eta, q, dip, nu):
    sd, cd, R, yt, dt = _common(xi, eta, q, dip)
    I1, I2, I3, I4, I5 = _I(xi, eta, q, dip, nu, R, yt, dt)
    at = _atan_term(xi, eta, q, R)
    ux = xi * q / (R * (R + eta)) + at + I1 * sd
    uy = yt * q / (R * (R + eta)) + q * cd / (R + eta) + I2 * sd
    uz = dt * q / (R * (R + eta)) + q * sd / (R + eta) + I4 * sd
    return np.array([ux, uy, uz])


def _ds(xi, eta, q, dip, nu):
    sd, cd, R, yt, dt = _common(xi, eta, q, dip)
    I1, I2, I3, I4, I5 = _I(xi, eta, q, dip, nu, R, yt, dt)
    at = _atan_term(xi, eta, q, R)
    ux = q / R - I3 * sd * cd
    uy = yt * q / (R * (R + xi)) + cd * at - I1 * sd * cd
    uz = dt * q / (R * (R + xi)) + sd * at - I5 * sd * cd
    return np.array([ux, uy, uz])


def _tf(xi, eta, q, dip, nu):
    sd, cd, R, yt, dt = _common(xi, eta, q, dip)
    I1, I2, I3, I4, I5 = _I(xi, eta, q, dip, nu, R, yt, dt)
    at = _atan_term(xi, eta, q, R)
    ux = q ** 2 / (R * (R + eta)) - I3 * sd ** 2
    uy = -dt * q / (R * (R + xi)) - sd * (xi * q / (R * (R + eta)) - at) - I1 * sd ** 2
    uz = yt * q / (R * (R + xi)) + cd * (xi * q / (R * (R + eta)) - at) - I5 * sd ** 2
    return np.array([ux, uy, uz])


def okada85_local(x, y, d, dip_deg, L, W, U1, U2, U3, nu=0.25):
    """Okada's own frame: x along strike from the fault corner, y horizontal normal to strike,
    d = depth of the BOTTOM edge, fault spans x in [0,L], up-dip width W.  -> (ux, uy, uz)"""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    dip = np.deg2rad(dip_deg)
    p = y * np.cos(dip) + d * np.sin(dip)
    q = y * np.sin(dip) - d * np.cos(dip)
    u = (-U1 / (2 * np.pi) * _chinnery(_ss, x, p, L, W, q, dip, nu)
         - U2 / (2 * np.pi) * _chinnery(_ds, x, p, L, W, q, dip, nu)
         + U3 / (2 * np.pi) * _chinnery(_tf, x, p, L, W, q, dip, nu))
    return u[0], u[1], u[2]


def rect_source(east, north, east_shift, north_shift, depth, strike, dip, rake, length, width,
                slip, opening_fraction=0.0, nu=0.25):
    """Surface displacement (ue, un, uz_up) [m] at points (east, north) [km] of a rectangular
    dislocation anchored at the centre of its top edge (see module docstring)."""
    east, north = np.asarray(east, dtype=np.float64), np.asarray(north, dtype=np.float64)
    st = np.deg2rad(strike)
    dp = np.deg2rad(dip)
    # bottom-edge depth and the horizontal offset of the bottom edge from the top edge
    d_bot = depth + width * np.sin(dp)
    # unit vectors: along strike (e,n) = (sin st, cos st); horizontal down-dip direction is
    # strike + 90 deg: (cos st, -sin st)
    ex, nx = np.sin(st), np.cos(st)
    ey, ny = np.cos(st), -np.sin(st)
    # Okada origin: surface projection of the bottom-left corner (x from 0..L along strike,
    # y positive in the direction where the fault comes UP, i.e. opposite to down-dip)
    oe = east_shift - 0.5 * length * ex + width * np.cos(dp) * ey
    on = north_shift - 0.5 * length * nx + width * np.cos(dp) * ny
    de, dn = east - oe, north - on
    x = de * ex + dn * nx
    y = -(de * ey + dn * ny)
    f = opening_fraction
    shear = slip * (1.0 - np.abs(f))
    U1 = shear * np.cos(np.deg2rad(rake))
    U2 = shear * np.sin(np.deg2rad(rake))
    U3 = slip * f
    ux, uy, uz = okada85_local(x, y, d_bot, dip, length, width, U1, U2, U3, nu)
    ue = ux * ex - uy * ey
    un = ux * nx - uy * ny
    return ue, un, uz


def mogi(east, north, east_shift, north_shift, depth, volume_change, nu=0.25):
    """Mogi (1958) point pressure source: (ue, un, uz_up) [m]; coordinates/depth [km],
    volume_change [m^3]"""
    de = (np.asarray(east, dtype=np.float64) - east_shift) * 1e3
    dn = (np.asarray(north, dtype=np.float64) - north_shift) * 1e3
    d = depth * 1e3
    R3 = (de ** 2 + dn ** 2 + d ** 2) ** 1.5
    c = (1.0 - nu) / np.pi * volume_change
    return c * de / R3, c * dn / R3, c * d / R3
