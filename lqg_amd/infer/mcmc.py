"""No-U-Turn sampler over the HIP likelihood — the role of `infer(..., method="nuts")` in lqg/infer/utils.py:14-41
(NumPyro NUTS over `lifted_model`: priors of lqg/infer/prior.py on the positive model parameters, likelihood =
System.log_likelihood of x[n, T, d] under model_type(T=T-1, **params), lqg/infer/models.py:20-34).

NumPyro is not part of this build; the sampler is restated here (Hoffman & Gelman 2014, Algorithm 6: efficient NUTS with
slice sampling and dual-averaging step-size adaptation) on the unconstrained parameters z = log(theta), identity mass
matrix.  What is MI355X-specific is how the gradient evaluations are issued: every chain is a coroutine that yields the
position it needs the log-density and gradient at; the driver collects ONE request per active chain and evaluates them
together as a CANDIDATE AXIS of the batched likelihood (one launch set for all chains — the reference runs chains on
separate XLA host devices, main.py:8, lqg/infer/utils.py:37).  Gradients: method "fd" (central differences in z, the
2P+1 candidates of every chain in the same sweep; works for every model shape) or "adjoint" (the reverse-mode HIP sweep
behind torch.autograd, lqg_amd/grad.py).  fp64 throughout.
"""
import math

import torch

from lqg_amd import options
from lqg_amd.infer import prior as _prior
from lqg_amd.infer.models import get_model_params


def log_prior(name, theta, prior_dict):
    """log density of the prior record of `name` at theta (elementwise), up to a constant."""
    rec = prior_dict[name]
    if rec[0] == "lognormal":
        return -torch.log(theta) - (torch.log(theta) - rec[1]) ** 2 / (2.0 * rec[2] ** 2)
    if rec[0] == "halfnormal":
        return -theta ** 2 / (2.0 * rec[1] ** 2)
    raise ValueError(f"unknown prior family {rec[0]!r} for {name}")


class Potential:
    """log p(x | exp z) + log prior(exp z) + sum z (Jacobian of the exp transform) and its gradient, for a [C, P] batch
    of positions (C = chains that asked in this round)."""

    def __init__(self, x, model_type, names, fixed, process_noise, dt, prior_dict, grad_method="fd", fd_step=1e-4, group=None):
        self.x, self.model_type, self.names, self.fixed = x.to(torch.float64), model_type, list(names), dict(fixed)
        self.pn, self.dt, self.prior, self.method, self.h, self.group = process_noise, dt, prior_dict, grad_method, fd_step, group
        self.evaluations = 0

    def _loglik(self, theta):                             # theta [C, P] -> [C] (differentiable when theta requires grad)
        from lqg_amd import _hip
        from lqg_amd import dist as ld
        x = self.x
        kw = dict(get_model_params(self.model_type))
        kw.update(self.fixed)
        kw.update({k: theta[:, i] for i, k in enumerate(self.names)})
        model = self.model_type(process_noise=self.pn, dt=self.dt, T=x.shape[-2] - 1, device=x.device, dtype=x.dtype, **kw)
        ll = model.log_likelihood(x)                      # [C, n]
        self.evaluations += theta.shape[0]
        if ll.requires_grad:
            return ll.sum(-1)
        return ld.all_reduce_sum(_hip.sum_trials(ll), group=self.group)

    def _graphed(self, z, C):
        """Captured finite-difference evaluator for up to the largest number of positions asked so far (fewer are padded);
        None when the evaluation cannot be captured or a process group shards the trials."""
        import os
        if self.group is not None or not z.is_cuda or not options.flag("GRAPH"):
            return None
        from lqg_amd.infer import graphed as _g
        if _g._sharded():
            return None
        cur = getattr(self, "_gev", None)
        if cur is None or (cur is not False and cur.K < C):
            from lqg_amd.infer import graphed
            ev = graphed.make(graphed.GraphedFiniteDifference, self.x, self.model_type, self.names, C, h=self.h,
                              fixed=self.fixed, process_noise=self.pn, dt=self.dt, extra=self.extra_and_grad)
            self._gev = ev if ev is not None else False
        return self._gev or None

    def _extra(self, z):                                  # prior + Jacobian, [C]
        theta = torch.exp(z)
        out = z.sum(-1)
        for i, k in enumerate(self.names):
            out = out + log_prior(k, theta[:, i], self.prior)
        return out

    def extra_and_grad(self, z):
        """_extra(z) and its gradient in closed form (z = log theta): lognormal(mu, s): -z - (z - mu)^2 / (2 s^2);
        halfnormal(s): -exp(2 z) / (2 s^2); Jacobian: + z.  Pure elementwise torch ops (capturable in the hipGraph)."""
        c = self.__dict__.get("_prior_consts")
        if c is None or c[0].device != z.device:
            ln = torch.tensor([1.0 if self.prior[k][0] == "lognormal" else 0.0 for k in self.names], dtype=torch.float64)
            mu = torch.tensor([float(self.prior[k][1]) if self.prior[k][0] == "lognormal" else 0.0 for k in self.names], dtype=torch.float64)
            i2 = torch.tensor([1.0 / float(self.prior[k][2 if self.prior[k][0] == "lognormal" else 1]) ** 2 for k in self.names],
                              dtype=torch.float64)
            for k in self.names:
                if self.prior[k][0] not in ("lognormal", "halfnormal"):
                    raise ValueError(f"unknown prior family {self.prior[k][0]!r} for {k}")
            c = self._prior_consts = tuple(t.to(z.device) for t in (ln, mu, i2))
        ln, mu, i2 = c
        e2 = torch.exp(2.0 * z)
        val = z + ln * (-z - 0.5 * (z - mu) ** 2 * i2) + (1.0 - ln) * (-0.5 * e2 * i2)
        grad = 1.0 + ln * (-1.0 - (z - mu) * i2) + (1.0 - ln) * (-e2 * i2)
        return val.sum(-1), grad

    def host(self, z):
        """z [C, P] on any device -> (logp [C], grad [C, P]) on the HOST with one device-to-host copy (what the chains'
        coroutines consume)."""
        z = z.to(device=self.x.device, dtype=torch.float64)
        C, P = z.shape
        if self.method == "fd":
            ev = self._graphed(z, C)
            if ev is not None:
                zp = z if C == ev.K else torch.cat([z, z[:1].expand(ev.K - C, P)])
                out = ev(zp)[:C].cpu()
                self.evaluations += C * (2 * P + 1)
                if not bool(torch.isnan(out).any()):      # (NaN: a frozen precondition failed at these values: eager path)
                    return out[:, 0], out[:, 1:]
                with options.override(GRAPH=0):
                    lp, gr = self(z)
                both = torch.cat([lp.detach()[:, None], gr.detach()], dim=1).cpu()
                return both[:, 0], both[:, 1:]
        lp, gr = self(z)
        both = torch.cat([lp.detach()[:, None], gr.detach()], dim=1).cpu()
        return both[:, 0], both[:, 1:]

    def __call__(self, z):
        from lqg_amd import dist as ld
        z = z.to(torch.float64)
        C, P = z.shape
        if self.method == "fd":
            ev = self._graphed(z, C)
            if ev is not None:      # the 2P+1 perturbed vectors of every position, model construction, sweeps, differences,
                zp = z if C == ev.K else torch.cat([z, z[:1].expand(ev.K - C, P)])     # prior: ONE hipGraph replay
                out = ev(zp)[:C].clone()                                                 # (infer/graphed.py; the graph owns its output)
                self.evaluations += C * (2 * P + 1)
                return out[:, 0], out[:, 1:]
        ex, eg = self.extra_and_grad(z)
        if self.method == "adjoint":
            zz = z.clone().requires_grad_(True)
            val = self._loglik(torch.exp(zz))             # this rank's trials
            val.sum().backward()
            both = ld.all_reduce_sum(torch.cat([val.detach()[:, None], zz.grad], dim=1), group=self.group)
            return both[:, 0] + ex, both[:, 1:] + eg
        eye = self.h * torch.eye(P, dtype=torch.float64, device=z.device)
        Z = torch.cat([z[:, None, :], z[:, None, :] + eye, z[:, None, :] - eye], dim=1).reshape(C * (2 * P + 1), P)
        with torch.no_grad():
            f = self._loglik(torch.exp(Z)).reshape(C, 2 * P + 1)
        return f[:, 0] + ex, (f[:, 1:P + 1] - f[:, P + 1:]) / (2 * self.h) + eg


def _finite(v):
    return v if math.isfinite(v) else -math.inf


def _leapfrog(z, r, g, eps):
    """One leapfrog step; coroutine: yields the new position, receives (logp, grad) there."""
    r = r + 0.5 * eps * g
    z = z + eps * r
    lp, g = yield z
    r = r + 0.5 * eps * g
    return z, r, _finite(lp), g


def _find_epsilon(z, lp, g, rng):
    """Heuristic initial step size (Hoffman & Gelman, Algorithm 4)."""
    eps = 0.1
    r = torch.randn(z.shape, generator=rng, dtype=torch.float64)
    joint0 = lp - 0.5 * float(r @ r)

    def energy(lp1, r1):
        j = lp1 - 0.5 * float(r1 @ r1)
        return j if math.isfinite(j) else -math.inf

    _, r1, lp1, _ = yield from _leapfrog(z, r, g, eps)
    joint = energy(lp1, r1)
    a = 1.0 if (joint - joint0) > math.log(0.5) else -1.0
    for _ in range(50):
        if not (a * (joint - joint0) > -a * math.log(2.0)):
            break
        eps *= 2.0 ** a
        _, r1, lp1, _ = yield from _leapfrog(z, r, g, eps)
        joint = energy(lp1, r1)
    return eps


def _build_tree(z, r, g, logu, v, j, eps, joint0, rng):
    """Recursive doubling (Algorithm 6); returns the usual tuple + the leaf statistics for dual averaging."""
    if j == 0:
        z1, r1, lp1, g1 = yield from _leapfrog(z, r, g, v * eps)
        joint = lp1 - 0.5 * float(r1 @ r1)
        joint = joint if math.isfinite(joint) else -math.inf
        n1 = 1 if logu <= joint else 0
        s1 = logu < joint + 1000.0
        alpha = min(1.0, math.exp(min(0.0, joint - joint0))) if math.isfinite(joint) else 0.0
        return z1, r1, g1, z1, r1, g1, z1, lp1, g1, n1, s1, alpha, 1, (not s1)
    zm, rm, gm, zp, rp, gp, z1, lp1, g1, n1, s1, a1, na1, div1 = yield from _build_tree(z, r, g, logu, v, j - 1, eps, joint0, rng)
    if s1:
        if v == -1:
            zm, rm, gm, _, _, _, z2, lp2, g2, n2, s2, a2, na2, div2 = yield from _build_tree(zm, rm, gm, logu, v, j - 1, eps, joint0, rng)
        else:
            _, _, _, zp, rp, gp, z2, lp2, g2, n2, s2, a2, na2, div2 = yield from _build_tree(zp, rp, gp, logu, v, j - 1, eps, joint0, rng)
        if n1 + n2 > 0 and float(torch.rand((), generator=rng)) < n2 / (n1 + n2):
            z1, lp1, g1 = z2, lp2, g2
        a1, na1, div1 = a1 + a2, na1 + na2, div1 or div2
        dz = zp - zm
        s1 = s2 and float(dz @ rm) >= 0 and float(dz @ rp) >= 0
        n1 = n1 + n2
    return zm, rm, gm, zp, rp, gp, z1, lp1, g1, n1, s1, a1, na1, div1


def nuts_chain(z0, num_warmup, num_samples, seed, max_depth=10, target_accept=0.8):
    """One chain as a coroutine: `z = next(gen)` / `z = gen.send((logp, grad))` ask for an evaluation at z (1-D tensor on
    the host); StopIteration.value = dict(samples [num_samples, P], accept, step_size, steps, divergences)."""
    rng = torch.Generator()
    rng.manual_seed(int(seed))
    z = z0.clone().to(torch.float64)
    lp, g = yield z
    lp = _finite(lp)
    eps = yield from _find_epsilon(z, lp, g, rng)
    mu, gamma, t0, kappa = math.log(10.0 * eps), 0.05, 10.0, 0.75
    hbar, log_eps_bar = 0.0, 0.0
    samples, accept, steps, ndiv = [], [], [], 0
    for it in range(num_warmup + num_samples):
        r0 = torch.randn(z.shape, generator=rng, dtype=torch.float64)
        joint0 = lp - 0.5 * float(r0 @ r0)
        logu = joint0 + math.log(max(float(torch.rand((), generator=rng)), 1e-300))
        zm = zp = z
        rm = rp = r0
        gm = gp = g
        j, n, s = 0, 1, True
        a_sum, n_a, n_leap = 0.0, 0, 0
        while s and j < max_depth:
            v = -1 if float(torch.rand((), generator=rng)) < 0.5 else 1
            if v == -1:
                zm, rm, gm, _, _, _, z1, lp1, g1, n1, s1, a1, na1, div = yield from _build_tree(zm, rm, gm, logu, v, j, eps, joint0, rng)
            else:
                _, _, _, zp, rp, gp, z1, lp1, g1, n1, s1, a1, na1, div = yield from _build_tree(zp, rp, gp, logu, v, j, eps, joint0, rng)
            if s1 and float(torch.rand((), generator=rng)) < min(1.0, n1 / n):
                z, lp, g = z1, lp1, g1
            n += n1
            dz = zp - zm
            s = s1 and float(dz @ rm) >= 0 and float(dz @ rp) >= 0
            a_sum, n_a, n_leap = a_sum + a1, n_a + na1, n_leap + na1
            ndiv += 1 if (div and it >= num_warmup) else 0
            j += 1
        acc = a_sum / max(n_a, 1)
        if it < num_warmup:                               # dual averaging (Algorithm 6, lines of Algorithm 5)
            m = it + 1
            hbar = (1 - 1 / (m + t0)) * hbar + (target_accept - acc) / (m + t0)
            log_eps = mu - math.sqrt(m) / gamma * hbar
            eta = m ** (-kappa)
            log_eps_bar = eta * log_eps + (1 - eta) * log_eps_bar
            eps = math.exp(log_eps)
            if m == num_warmup:
                eps = math.exp(log_eps_bar)
        else:
            samples.append(z.clone())
            accept.append(acc)
            steps.append(n_leap)
    return dict(samples=torch.stack(samples) if samples else torch.empty(0, z.numel(), dtype=torch.float64),
                accept=accept, step_size=eps, steps=steps, divergences=ndiv)


def run_chains(potential, z0, num_warmup, num_samples, seed=0, max_depth=10, target_accept=0.8):
    """Drive len(z0) chains in lock step: one batched evaluation of `potential(Z[C', P]) -> (logp[C'], grad[C', P])` per
    round, C' = chains still running.  Returns the per-chain result dicts."""
    gens = [nuts_chain(z, num_warmup, num_samples, seed + 7919 * c, max_depth, target_accept) for c, z in enumerate(z0)]
    req = [next(g) for g in gens]
    done = [None] * len(gens)
    active = list(range(len(gens)))
    dev = getattr(potential, "x", torch.zeros(())).device
    while active:
        Z = torch.stack([req[c] for c in active])
        if hasattr(potential, "host"):
            lp, gr = potential.host(Z)
        else:
            lp, gr = potential(Z.to(dev))
            lp, gr = lp.detach().cpu(), gr.detach().cpu()
        still = []
        for k, c in enumerate(active):
            g_ = torch.nan_to_num(gr[k], nan=0.0, posinf=0.0, neginf=0.0)
            try:
                req[c] = gens[c].send((float(lp[k]), g_))
                still.append(c)
            except StopIteration as stop:
                done[c] = stop.value
        active = still
    return done


class MCMCResult:
    """What the reference's callers use of numpyro's MCMC object: get_samples(group_by_chain), print_summary."""

    def __init__(self, names, chains):
        self.names, self.chains = list(names), chains
        self.num_chains = len(chains)

    def get_samples(self, group_by_chain=False):
        out = {}
        for i, k in enumerate(self.names):
            per = torch.stack([torch.exp(c["samples"][:, i]) for c in self.chains])          # [chains, samples]
            out[k] = per if group_by_chain else per.reshape(-1)
        return out

    def get_extra_fields(self):
        return dict(accept_prob=torch.tensor([c["accept"] for c in self.chains]),
                    num_steps=torch.tensor([c["steps"] for c in self.chains]),
                    step_size=torch.tensor([c["step_size"] for c in self.chains]),
                    diverging=torch.tensor([c["divergences"] for c in self.chains]))

    def summary(self):
        rows = {}
        for k, v in self.get_samples(group_by_chain=True).items():
            flat = v.reshape(-1)
            q = torch.quantile(flat, torch.tensor([0.05, 0.5, 0.95], dtype=flat.dtype))
            r_hat = float("nan")
            if v.shape[0] > 1 and v.shape[1] > 1:           # Gelman-Rubin
                n = v.shape[1]
                W = v.var(dim=1, unbiased=True).mean()
                B = n * v.mean(dim=1).var(unbiased=True)
                r_hat = float(torch.sqrt(((n - 1) / n * W + B / n) / W))
            rows[k] = dict(mean=float(flat.mean()), std=float(flat.std()), median=float(q[1]), q5=float(q[0]),
                           q95=float(q[2]), r_hat=r_hat)
        return rows

    def print_summary(self):
        print(f"{'':>20s} {'mean':>10s} {'std':>10s} {'median':>10s} {'5.0%':>10s} {'95.0%':>10s} {'r_hat':>8s}")
        for k, r in self.summary().items():
            print(f"{k:>20s} {r['mean']:10.3f} {r['std']:10.3f} {r['median']:10.3f} {r['q5']:10.3f} {r['q95']:10.3f} {r['r_hat']:8.3f}")


def infer_nuts(x, num_samples, num_warmup, model, process_noise=1.0, dt=1.0 / 60, num_chains=1, seed=0, grad_method="fd",
               prior_dict=None, max_depth=10, target_accept=0.8, group=None, **fixed):
    prior_dict = _prior.default_prior if prior_dict is None else prior_dict
    names = [k for k in get_model_params(model) if k not in fixed]
    pot = Potential(x, model, names, fixed, process_noise, dt, prior_dict, grad_method=grad_method, group=group)
    # init_to_median of the reference (lqg/infer/utils.py:18): prior medians, jittered per chain
    g = torch.Generator()
    g.manual_seed(int(seed))
    med = []
    for k in names:
        rec = prior_dict[k]
        med.append(rec[1] if rec[0] == "lognormal" else math.log(rec[1] * 0.6744897501960817))
    z0 = [torch.tensor(med, dtype=torch.float64) + (0.0 if c == 0 else 0.1) * torch.randn(len(names), generator=g, dtype=torch.float64)
          for c in range(num_chains)]
    res = MCMCResult(names, run_chains(pot, z0, num_warmup, num_samples, seed=seed, max_depth=max_depth,
                                       target_accept=target_accept))
    res.evaluations = pot.evaluations
    return res
