"""NumPy statement of BASELINE configs[0] in the shape of the reference's CPU path
(ShanChen2D/SimpleD2Q9.py): whole-lattice arrays `[9, ny, nx]`, the step as macroscopic sums ->
equilibrium -> BGK -> bounce-back by mask -> `np.roll` streaming (SimpleD2Q9.py:170-241, :302-321),
extended by the two-component Shan-Chen interaction of the original GPU kernel
(OptimizedD2Q9GPU.py:1274-1449: force from psi products, velocity shift tau F / rho on the common
velocity), because SimpleD2Q9 itself is single-phase and its loop does not run (undefined names in
:173-176, :228, :266).  Fully periodic box, psi = rho.

TEST INFRASTRUCTURE / CPU BASELINE ONLY (rules in oracle/__init__.py).  Pinned transitively: equal
to oracle/sc_oracle.c with the boundary kernels skipped (tests/test_oracle_sc.py), which is pinned
to the real reference driver.  Single-threaded NumPy: this is the "repo's own CPU path" line of
bench.py.
"""
import numpy as np

EX = np.array([0, 1, 0, -1, 0, 1, -1, -1, 1])
EY = np.array([0, 0, 1, 0, -1, 1, 1, -1, -1])
W = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
OPP = [0, 3, 4, 1, 2, 7, 8, 5, 6]
WI = np.array([1. / 9.] * 4 + [1. / 36.] * 4)          # interaction weights of directions 1..8


class SimpleD2Q9SC:
    def __init__(self, is_domain, rho0, rho1, tau=(1.0, 1.0), G=3.8, Gs=(-0.40, 0.40)):
        self.isDomain = np.asarray(is_domain) == 1
        self.isWall = ~self.isDomain
        self.tau = (float(tau[0]), float(tau[1]))
        self.G = np.array([[0.0, G], [G, 0.0]])          # interCoeff, ShanChenD2Q9.py:375-383
        self.Gs = (float(Gs[0]), float(Gs[1]))
        self.particleDisFunc = [W[:, None, None] * np.where(self.isDomain, r, 0.0)[None] for r in (rho0, rho1)]
        self.rho = [f.sum(axis=0) for f in self.particleDisFunc]

    @staticmethod
    def _from(a, i):
        """value of `a` at the neighbour in direction i (periodic)"""
        return np.roll(np.roll(a, -EY[i], axis=0), -EX[i], axis=1)

    def step(self):
        f, tau = self.particleDisFunc, self.tau
        # macroscopic parameters (SimpleD2Q9.py:170-176) and the common velocity of O:1283-1292
        self.rho = [g.sum(axis=0) for g in f]
        rho = self.rho
        mom = [((g * EX[:, None, None]).sum(axis=0), (g * EY[:, None, None]).sum(axis=0)) for g in f]
        with np.errstate(divide="ignore", invalid="ignore"):
            rt = rho[0] / tau[0] + rho[1] / tau[1]
            pvx = (mom[0][0] / tau[0] + mom[1][0] / tau[1]) / rt
            pvy = (mom[0][1] / tau[0] + mom[1][1] / tau[1]) / rt
            out = []
            for k in range(2):
                # interaction force: fluid neighbours through psi products, solid neighbours through Gs (O:1293-1345)
                fx = np.zeros_like(rho[k]); fy = np.zeros_like(rho[k])
                for i in range(1, 9):
                    fluid_nb = self._from(self.isDomain, i)
                    for j in range(2):
                        t = -WI[i - 1] * self.G[k, j] * rho[k] * self._from(rho[j], i)
                        fx += np.where(fluid_nb, t * EX[i], 0.0); fy += np.where(fluid_nb, t * EY[i], 0.0)
                    s = -WI[i - 1] * self.Gs[k] * rho[k]
                    fx += np.where(fluid_nb, 0.0, s * EX[i]); fy += np.where(fluid_nb, 0.0, s * EY[i])
                ux = pvx + tau[k] * fx / rho[k]; uy = pvy + tau[k] * fy / rho[k]
                usq = ux * ux + uy * uy
                # equilibrium + BGK (SimpleD2Q9.py:178-216)
                eu = EX[:, None, None] * ux[None] + EY[:, None, None] * uy[None]
                feq = W[:, None, None] * rho[k][None] * (1.0 + 3.0 * eu + 4.5 * eu * eu - 1.5 * usq[None])
                post = f[k] - (f[k] - feq) / tau[k]
                out.append(np.where(self.isDomain[None], post, 0.0))
        # streaming by np.roll (SimpleD2Q9.py:232-241) with half-way bounce-back at walls (:220-230)
        for k in range(2):
            new = np.empty_like(out[k])
            for i in range(9):
                moved = np.roll(np.roll(out[k][i], EY[i], axis=0), EX[i], axis=1)
                from_wall = np.roll(np.roll(self.isWall, EY[i], axis=0), EX[i], axis=1)
                new[i] = np.where(from_wall, out[k][OPP[i]], moved)
            self.particleDisFunc[k] = np.where(self.isDomain[None], new, 0.0)

    def run(self, nsteps):
        for _ in range(int(nsteps)):
            self.step()
        self.rho = [g.sum(axis=0) for g in self.particleDisFunc]
        return self
