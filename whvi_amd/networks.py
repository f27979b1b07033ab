"""WHVI networks (mirror of the reference's src/networks.py): the Monte-Carlo sample loop, the
ELBO, the two-phase training loop and evaluation.  Consumers of ``WHVILinear``; kept
interface- and checkpoint-compatible (state_dict keys ``sequential.{i}....`` and
``likelihood.sigma``)."""
import pathlib
from typing import Iterable, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from whvi_amd.layers import WHVI
from whvi_amd.likelihoods import GaussianLikelihood, Likelihood

__all__ = ["WHVINetwork", "WHVIRegression"]


def _progress(iterable, desc):
    try:
        from tqdm import tqdm
        return tqdm(iterable, desc=desc)
    except ImportError:  # tqdm is optional here
        class _Plain:
            def __init__(self, it):
                self.it = it

            def __iter__(self):
                return iter(self.it)

            def set_description(self, *_):
                pass
        return _Plain(iterable)


class WHVINetwork(nn.Module, WHVI):
    def __init__(self, modules: Iterable[nn.Module], likelihood: Likelihood, train_samples=1, eval_samples=64):
        """Sequential network with WHVI layers (src/networks.py:12-31)."""
        super().__init__()
        self.sequential = nn.Sequential(*modules)
        self.likelihood = likelihood
        self.train_samples = train_samples
        self.eval_samples = eval_samples
        self.current_mnll = 0.0
        self.current_kl = 0.0
        # "loop": the reference's per-sample Python loop (src/networks.py:47-51), same RNG stream;
        # "batched": every WHVI layer draws and applies all samples at once (one fused launch + one
        # batched GEMM per layer); "auto" = batched on the GPU, loop on the host.
        self.mc_mode = "auto"

    @property
    def kl(self):
        return sum([m.kl for m in self.sequential.children() if 'kl' in dir(m)])

    def pack_parameters(self):
        """Opt in to the packed parameter layout of every stacked layer (``WHVIStackedMatrix.pack_parameters``):
        ``4 * stack`` parameter tensors per layer become 4, checkpoints keep the reference's keys.  Call it before
        creating the optimizer."""
        for module in self.modules():
            if hasattr(module, "pack_parameters") and module is not self:
                module.pack_parameters()
        return self

    def set_faithful_dataflow(self, on: bool = True):
        """``on``: every WHVI layer of the network evaluates the reference's dataflow op for op on the GPU too -- square layers
        build their (as written, exactly diagonal) weight matrices and multiply with a dense GEMM, stacked and column layers call
        ``torch.matmul`` -- instead of the shipped one-launch routes (``whvi_diag_apply``, ``whvi_small_k_apply``,
        ``whvi_row_dot``), which return the same values.  For cross-checks and A/B timings (bench.py prints both)."""
        for module in self.modules():
            if hasattr(type(module), "faithful_dataflow"):
                module.faithful_dataflow = bool(on)
            if hasattr(type(module), "hip_apply"):
                module.hip_apply = not on
        return self

    def set_inkernel_rng(self, on: bool = True):
        """Opt in to drawing eps inside the reparameterisation kernel for the batched MC passes on the GPU
        (``whvi_reparam_kl_philox_f32``): one launch less per layer and pass, hipGraph-safe.  The
        numbers then come from this library's Philox stream, not from ``torch.randn`` (seeded from torch's default
        generator when a layer first draws); the per-sample loop mode is unaffected."""
        for module in self.modules():
            if hasattr(type(module), "inkernel_rng"):
                module.inkernel_rng = bool(on)
                module._rng_state = None
        return self

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(batch, in_dim) -> (batch, out_dim, n_samples): one stochastic pass per Monte-Carlo
        sample, samples stacked on the last axis (src/networks.py:36-54)."""
        assert x.dim() == 2, "Input shape must be (batch_size, in_dim)"
        batch_size = x.size(0)
        n_samples = self.train_samples if self.training else self.eval_samples
        mode = self.mc_mode
        if mode == "auto":
            mode = "batched" if x.device.type == "cuda" else "loop"
        if mode == "batched":
            return self.forward_batched(x, n_samples)
        draws = []
        for _ in range(n_samples):
            out = self.sequential.forward(x)
            draws.append(out.reshape(batch_size, out.size(-1)))
        predictions = torch.stack(draws, dim=2)
        assert predictions.dim() == 3
        return predictions

    def forward_batched(self, x: torch.Tensor, n_samples: int) -> torch.Tensor:
        """All Monte-Carlo samples in one pass.  Activations carry a leading sample
        axis ``(S, batch, features)`` from the first WHVI layer on; deterministic modules broadcast
        over it.  Same output layout as the loop: ``(batch, out_dim, n_samples)``."""
        h = x
        fused_kl, complete = None, True
        modules = list(self.sequential)
        i = 0
        while i < len(modules):
            module = modules[i]
            # an nn.ReLU whose neighbour is a layer that folds activations into its own launch (the square layer's
            # whvi_diag_apply on the GPU) is not run as a pass of its own: one read + one write of the (S, batch, D)
            # activations instead of three.  Same values: the kernel applies max(., 0) on load / before the store.
            relu_in = False
            if type(module) is nn.ReLU and i + 1 < len(modules) and _fuses_relu(modules[i + 1], h):
                relu_in, i = True, i + 1
                module = modules[i]
            if hasattr(module, "forward_mc"):
                relu_out = (i + 1 < len(modules) and type(modules[i + 1]) is nn.ReLU and _fuses_relu(module, h))
                if relu_in or relu_out:
                    h = module.forward_mc(h, n_samples, relu_in=relu_in, relu_out=relu_out)
                    i += 1 if relu_out else 0
                else:
                    h = module.forward_mc(h, n_samples)
                kl = getattr(module, "_mc_kl", None)
                module._mc_kl = None
                if kl is None:
                    complete = False
                else:
                    fused_kl = kl if fused_kl is None else fused_kl + kl
            else:
                h = module(h)
                complete = complete and 'kl' not in dir(module)
            i += 1
        # sum of the layers' KL terms computed inside this very pass (fused kernel, same parameter values,
        # same autograd graph); consumed -- once -- by loss()
        self._pass_kl = fused_kl if complete and torch.is_tensor(fused_kl) else None
        if h.dim() == 2:                      # no stochastic layer at all: identical samples
            h = h.unsqueeze(0).expand(n_samples, *h.shape)
        return h.permute(1, 2, 0)

    def loss(self, x, y, n: int, ignore_kl=False) -> torch.Tensor:
        """Negative ELBO estimate = MNLL (+ KL) (src/networks.py:56-69)."""
        self._pass_kl = None
        mnll = self.likelihood.mnll_batch_estimate(y, self(x), n)
        pass_kl, self._pass_kl = getattr(self, "_pass_kl", None), None
        kl = pass_kl if (pass_kl is not None and not ignore_kl) else self.kl
        # the monitoring attributes of src/networks.py:66-67, kept as detached values: holding the graph tensors
        # here would keep the whole autograd graph (and its gradient accumulators) alive until the next call
        self.current_mnll = mnll.detach() if torch.is_tensor(mnll) else mnll
        self.current_kl = kl.detach() if torch.is_tensor(kl) else kl
        return mnll if ignore_kl else mnll + kl

    def _epochs(self, data_loader, optimizer, scheduler, epochs, label, ignore_kl, pbar_update_period,
                checkpoint_dir=None, set_to_none=False, graphed=None, sharded=False):
        bar = _progress(range(epochs), f'[{label}] KL = {self.current_kl:.2f}, MNLL = {self.current_mnll:.2f}')
        n = len(data_loader.dataset)
        for epoch in bar:
            for data_x, data_y in data_loader:
                if sharded:
                    # MC samples sharded over the ranks of the process group: every rank runs the same batch for its
                    # share of the samples, one all-reduce of the parameter gradients, identical steps everywhere
                    from whvi_amd import parallel
                    self._sharded_steps = getattr(self, "_sharded_steps", 0) + 1
                    parallel.mc_sharded_loss(self, data_x, data_y, n, self.train_samples, base_seed=self._sharded_steps,
                                             ignore_kl=ignore_kl)
                    optimizer.step()
                    scheduler.step()
                    self.zero_grad(set_to_none=set_to_none)
                    continue
                if graphed is not None:
                    # the whole step -- loss, backward, optimizer.step(), scheduler.step() -- as one hipGraph replay, one
                    # captured step per batch shape (a data set whose size is no multiple of the batch size has two: the
                    # full batch and the short last one); any further shape takes the eager step
                    step = self._graphed_step_for(graphed, data_x, data_y, n, optimizer, scheduler, ignore_kl)
                    if step is not None:
                        hook = graphed.get("before_replay")
                        if hook is not None:
                            hook(step)
                        step(data_x, data_y)
                        continue
                    # eager step beside a captured one: .grad still names a graph's static buffers -- detach from them
                    # first, or this backward would accumulate into what the last replay left there
                    self.zero_grad(set_to_none=True)
                loss = self.loss(data_x, data_y, n=n, ignore_kl=ignore_kl)
                loss.backward()
                del loss                      # no graph of this pass may outlive it (GraphedTrainStep's precondition)
                optimizer.step()
                scheduler.step()
                self.zero_grad(set_to_none=set_to_none if graphed is None else True)
            if checkpoint_dir is not None and epoch % 5000 == 0 and (not sharded or self._is_rank_zero()):
                torch.save(self.state_dict(), pathlib.Path(checkpoint_dir) / f'epoch-{epoch}.pth')
            if epoch % pbar_update_period == 0:
                bar.set_description(f'[{label}] KL = {self.current_kl:.2f}, MNLL = {self.current_mnll:.2f}')

    def _graphed_step_for(self, graphed, data_x, data_y, n, optimizer, scheduler, ignore_kl):
        """The captured step for this batch shape: from this ``train_model`` call, from an earlier one on the same
        optimizer / schedule / data-set size (``self._train_graphs``: a later call continues on the same graphs), or
        captured now -- up to ``max_shapes`` of them; ``None`` = take the eager step."""
        key = (tuple(data_x.shape), tuple(data_y.shape), data_x.dtype, bool(ignore_kl))
        step = graphed["steps"].get(key)
        if step is not None:
            return step
        from whvi_amd.graphs import GraphedTrainStep
        options = graphed["options"]
        kept = getattr(self, "_train_graphs", None)
        if kept is None or kept["optimizer"] is not optimizer or kept["scheduler"] is not scheduler or kept["n"] != n \
                or kept["static_eps"] != bool(options.get("static_eps", False)):
            kept = self._train_graphs = {"optimizer": optimizer, "scheduler": scheduler, "n": n, "steps": {},
                                         "static_eps": bool(options.get("static_eps", False))}
        step = kept["steps"].get(key)
        if step is not None:
            step.restore_static_eps()
        elif sum(1 for k in kept["steps"] if k[3] == key[3]) < graphed["max_shapes"]:
            for other in kept["steps"].values():
                other.restore_static_eps()                   # a new capture shares the static eps buffers of the others
            step = kept["steps"][key] = GraphedTrainStep(self, optimizer, data_x, data_y, n, ignore_kl=ignore_kl,
                                                         scheduler=scheduler, **options)
        else:
            return None
        graphed["steps"][key] = step
        if graphed.get("first") is None:
            graphed["first"] = step
        return step

    @staticmethod
    def _is_rank_zero():
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0

    def train_model(self, data_loader, optimizer, scheduler, epochs1: int = 500, epochs2: int = 5000,
                    pbar_update_period=20, ignore_kl=False, checkpoint_dir=None, graphed=False, packed=None,
                    graph_options=None, sharded=None):
        """Two phases as in src/networks.py:71-99.  As in the reference, the ``requires_grad``
        assignments below set a plain attribute on the likelihood MODULE and do not freeze its
        ``sigma`` parameter -- kept so that training trajectories agree.

        ``graphed=True`` (GPU; not in the reference) runs the SAME recipe -- the two phases, ``scheduler.step()`` after
        every batch, the checkpoint cadence, the reference's ``state_dict`` keys -- with every step a hipGraph replay
        (``whvi_amd.graphs.GraphedTrainStep``).  It needs an optimizer and a schedule whose state lives on the device:
        ``make_optimizer(net, capturable=True[, packed=True])``.  ``packed=True`` additionally insists that the stacked
        layers use the packed parameter layout (it has to be chosen BEFORE the optimizer is created: the optimizer holds
        the parameter tensors).  ``graph_options``: keyword arguments for ``GraphedTrainStep`` (``static_eps``, ``warmup``)
        plus an optional ``before_replay(step)`` callable run ahead of every replay and ``max_shapes`` (default 2): how
        many batch shapes get a captured step of their own -- the full batch and a short last one; further shapes take the
        eager step.  Returns the captured step of the first batch shape (``None`` when not graphed).

        ``sharded`` (not in the reference, which is single-process): ``True`` / ``False`` to force; ``None`` = automatically,
        and only where sharding cannot change what an existing caller gets: inside an initialised ``torch.distributed``
        process group, NOT graphed, and with at least one Monte-Carlo sample per rank (``train_samples >= world`` -- with
        the default ``train_samples = 1`` a data-parallel job that feeds every rank its own batches keeps doing exactly
        that).  The ``train_samples`` Monte-Carlo samples of every
        step are then split over the ranks (``whvi_amd.parallel.mc_sharded_loss``): each rank runs the SAME batch --
        feed every rank the same data -- for its share of the samples, gradients are summed in one all-reduce, and the
        replicated parameters stay bit-equal across ranks.  Step k draws from generators seeded with (k, rank)."""
        if sharded is None:
            from whvi_amd import parallel
            sharded = (not graphed) and parallel._in_group() and self.train_samples >= parallel._world()[1]
        if sharded and graphed:
            raise RuntimeError("train_model: graphed=True and sharded=True cannot be combined (the gradient all-reduce is "
                               "not part of the captured step)")
        if packed:
            unpacked = [m for m in self.modules() if hasattr(m, "pack_parameters") and m is not self and not m._packed]
            if unpacked:
                raise RuntimeError("train_model(packed=True): call net.pack_parameters() -- or make_optimizer(net, packed=True) "
                                   "-- BEFORE creating the optimizer; the optimizer holds the parameter tensors")
        state = None
        if graphed:
            options = dict(graph_options or {})
            state = {"steps": {}, "first": None, "before_replay": options.pop("before_replay", None),
                     "max_shapes": int(options.pop("max_shapes", 2)), "options": options}
        self.train()
        self.likelihood.requires_grad = False
        self._epochs(data_loader, optimizer, scheduler, epochs1, 'Fixed LH', ignore_kl, pbar_update_period,
                     set_to_none=True, graphed=state, sharded=sharded)
        self.likelihood.requires_grad = True
        self._epochs(data_loader, optimizer, scheduler, epochs2, 'Optimized LH', ignore_kl, pbar_update_period,
                     checkpoint_dir=checkpoint_dir, graphed=state, sharded=sharded)
        if state is not None:
            for step in state["steps"].values():
                step.release_static_eps()
        self.eval()
        return state["first"] if state is not None else None

    def eval_model(self, X_test, y_test, loss) -> Tuple[float, float]:
        """(test error, test MNLL) with ``eval_samples`` draws (src/networks.py:101-115)."""
        self.eval()
        y_pred = self(X_test)
        test_mnll = self.likelihood.mnll_batch_estimate(y_test, y_pred, n=y_test.size(0))
        return float(loss(y_pred, y_test).detach()), float(test_mnll.detach())


def _fuses_relu(module, h):
    fn = getattr(module, "fuses_relu", None)
    return bool(fn is not None and hasattr(module, "forward_mc") and h.device.type == "cuda" and fn(h))


def _rmse_of_mean(y_pred, y_true):
    return torch.sqrt(F.mse_loss(y_pred.mean(dim=2).flatten(), y_true.flatten()))


class WHVIRegression(WHVINetwork):
    def __init__(self, modules: Iterable[nn.Module], sigma: float = 1.0, **kwargs):
        """Regression network with a Gaussian likelihood (src/networks.py:118-128)."""
        super().__init__(modules, likelihood=GaussianLikelihood(sigma), **kwargs)

    def eval_model(self, X_test, y_test, loss=_rmse_of_mean) -> Tuple[float, float]:
        return super().eval_model(X_test, y_test, loss)
