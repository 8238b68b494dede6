"""Independent numpy evaluation of the factor step and of the per-step affine terms, node by node, from the reference's formulas:
Engine::initialiseSystemDevice / factorStep (/root/reference/src/Engine.cu:382-463, 671-774; preconditioning kernels
Utilities.cu:33-58, 360-405) and Engine::eliminateInputDistubanceCoupling (Engine.cu:1147-1298; calculateDiffUhat / calculateZeta,
Utilities.cu:69-131).  Unlike the CPU oracle it never materialises per-node blocks for the whole tree, so it can check single
nodes of a tree whose blocks do not fit the host (wide4096: 160 GB) -- tests/test_numpy_engine.py pins it to the oracle on small
trees, tests/test_gpu_baseline_configs.py uses it on the HIP path at full size.  Test infrastructure (like oracle/)."""
import numpy as np


class NumpyEngine:
    def __init__(self, network, tree, config):
        g = lambda d, k: int(np.asarray(d[k]).ravel()[0])
        self.nx, self.nu, self.nd = g(network, "nx"), g(network, "nu"), g(network, "nd")
        self.nv, self.N, self.nodes = g(config, "nv"), g(tree, "N"), g(tree, "nodes")
        nx, nu, nd, nv, N = self.nx, self.nu, self.nd, self.nv, self.N
        col = lambda v, r, c: np.asarray(v, float).reshape(r, c, order="F")
        self.B, self.Gd = col(network["matB"], nx, nu), col(network["matGd"], nx, nd)
        self.L, self.Lhat, self.W = col(config["matL"], nu, nv), col(config["matLhat"], nu, nd), col(config["costW"], nu, nu)
        self.diag = np.asarray(config["matDiagPrecnd"], float).reshape(N, 2 * nx + nu)      # per stage: d_u | d_x | d_xs
        self.stage = np.asarray(tree["stages"], int)
        self.anc = np.asarray(tree["ancestor"], int) - 1
        self.p = np.asarray(tree["probNode"], float)
        self.errD = np.asarray(tree["errorDemandNode"], float).reshape(self.nodes, nd)
        self.errP = np.asarray(tree["errorPriceNode"], float).reshape(self.nodes, nu)
        self.alpha1 = np.asarray(network["costAlpha1"], float)
        self.prevDemand = np.asarray(config["prevDemand"], float)
        self.bounds = {k: np.asarray(network[k], float) for k in ("vecXmin", "vecXmax", "vecXsafe", "vecUmin", "vecUmax")}
        self.Rinv = np.linalg.inv(self.L.T @ self.W @ self.L)      # Omega_i = Rinv / p_i (Engine.cu:412-416, 707-714)
        self.Bbt = self.L.T @ self.B.T                              # Gtil = (B L)' (Engine.cu:702-705)

    # ---- factor step, one node: blocks in the reference's layout (column-major, nv rows) ---------------------------------
    def operators(self, node):
        """{"Phi": nv x 2nx, "D": nv x 2nx, "Psi": nv x nu, "Ftil": nv x nu}: F_i = sqrt(p_i) [diag(d_x); diag(d_xs)], G_i =
        sqrt(p_i) diag(d_u), D_i = Gtil F_i', Ftil_i = L' G_i', Phi_i = -Omega_i D_i / 2, Psi_i = -Omega_i Ftil_i / 2 (Engine.cu:721-745)"""
        nx, nu = self.nx, self.nu
        sp = np.sqrt(self.p[node])
        dk = self.diag[self.stage[node]]
        F = sp * np.vstack([np.diag(dk[nu:nu + nx]), np.diag(dk[nu + nx:])])       # 2nx x nx
        G = sp * np.diag(dk[:nu])
        D = self.Bbt @ F.T
        Ft = self.L.T @ G.T
        Om = self.Rinv / self.p[node]
        return {"Phi": -0.5 * Om @ D, "D": D, "Psi": -0.5 * Om @ Ft, "Ftil": Ft}

    # ---- scaled bounds of a set of nodes (preconditionConstraintX / U, Utilities.cu:360-405) -------------------------------
    def bounds_of(self, nodes):
        nx, nu = self.nx, self.nu
        nodes = np.asarray(nodes, int)
        sp = np.sqrt(self.p[nodes])[:, None]
        dk = self.diag[self.stage[nodes]]
        b = self.bounds
        return {"xmin": sp * dk[:, nu:nu + nx] * b["vecXmin"], "xmax": sp * dk[:, nu:nu + nx] * b["vecXmax"],
                "xs": sp * dk[:, nu + nx:] * b["vecXsafe"], "umin": sp * dk[:, :nu] * b["vecUmin"], "umax": sp * dk[:, :nu] * b["vecUmax"]}

    # ---- affine terms of the whole tree (vectors only) -----------------------------------------------------------------------
    def affine(self, dhat, ahat, w_eco=1.0):
        """uhat_i = Lhat d_i, e_i = Gd d_i with d_i = eps_d_i + dhat[stage]; alpha_i = w_e (eps_a_i + ahat[stage] + alpha1);
        zeta_i = p_i (uhat_i - uhat_anc) - sum_children p_c (uhat_c - uhat_i), root against Lhat prevDemand;
        beta_i = 2 (W L)' zeta_i + p_i L' alpha_i"""
        nd, nu = self.nd, self.nu
        d = self.errD + np.asarray(dhat, float).reshape(-1, nd)[self.stage]
        uhat = d @ self.Lhat.T
        e = d @ self.Gd.T
        alpha = w_eco * (self.errP + np.asarray(ahat, float).reshape(-1, nu)[self.stage] + self.alpha1)
        prev_uhat = self.Lhat @ self.prevDemand
        du = uhat - np.vstack([prev_uhat[None, :], uhat[self.anc[1:]]])
        zeta = self.p[:, None] * du
        np.subtract.at(zeta, self.anc[1:], self.p[1:, None] * du[1:])
        beta = 2.0 * zeta @ (self.W @ self.L) + self.p[:, None] * (alpha @ self.L)
        return {"uhat": uhat, "e": e, "alpha": alpha, "beta": beta}
