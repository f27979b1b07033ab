"""hipGraph replay of the launch-bound predictive pass.

A small WHVI network's Monte-Carlo forward is a few dozen tiny kernels (per layer: randn, softplus, the
fused weight launch, a batched GEMM, ...), i.e. bound by launch latency, not by the GPU.  The C ABI is
capture-safe (no allocation, no synchronisation, launches on the stream it is given), so the whole batched
pass can be recorded once into a hipGraph (``torch.cuda.graph``) and replayed: one launch per forward.
torch's generator is graph-aware, so every replay draws fresh eps.

``GraphedTrainStep`` does the same for a whole training step (loss, backward, optimizer, learning-rate schedule): the
backward of the fused weight kernel runs from the autograd engine's thread and is captured with the rest
(tools/graph_train_probe.py walks through the stages); ``WHVINetwork.train_model(..., graphed=True)`` runs the
reference's two-phase recipe on it.
"""
import gc
import warnings

import torch

__all__ = ["GraphedPredictor", "GraphedTrainStep"]


class GraphedPredictor:
    """``pred = GraphedPredictor(net, example_x, n_samples)(x)`` == ``net.forward_batched(x, n_samples)``
    under ``torch.no_grad()``, replayed from a hipGraph.  ``x`` must keep the example's shape and dtype; it
    is copied into a static buffer and the returned tensor is a static buffer too (clone it to keep it)."""

    def __init__(self, net, example_x: torch.Tensor, n_samples: int = None, warmup: int = 3):
        if example_x.device.type != "cuda":
            raise RuntimeError("GraphedPredictor needs a GPU tensor")
        self.net = net
        self.n_samples = int(n_samples if n_samples is not None else net.eval_samples)
        self.static_x = example_x.detach().clone()
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=example_x.device)
        with torch.no_grad():
            side.wait_stream(torch.cuda.current_stream(example_x.device))
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    net.forward_batched(self.static_x, self.n_samples)
            torch.cuda.current_stream(example_x.device).wait_stream(side)
            with torch.cuda.graph(self.graph):
                self.static_out = net.forward_batched(self.static_x, self.n_samples)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape != self.static_x.shape or x.dtype != self.static_x.dtype:
            raise RuntimeError("GraphedPredictor: input shape/dtype differs from the captured example")
        self.static_x.copy_(x)
        self.graph.replay()
        return self.static_out


def _reaches(root, wanted, limit=200_000):
    """True when the autograd graph under ``root`` (a grad_fn) contains the gradient accumulator of one of ``wanted``
    (a set of parameter ids)."""
    seen, stack = set(), [root]
    while stack and len(seen) < limit:
        node = stack.pop()
        if node is None or id(node) in seen:
            continue
        seen.add(id(node))
        variable = getattr(node, "variable", None)                 # AccumulateGrad nodes carry their leaf
        if variable is not None and id(variable) in wanted:
            return True
        stack.extend(fn for fn, _ in node.next_functions)
    return False


def _stale_graph_holders(params, device):
    """Live tensors on ``device`` that still carry an autograd graph reaching ``params`` (a kept ``loss``, a list of
    losses, ...).  Text-independent companion of the warning check below: found by walking the garbage collector's
    tensors, so it does not depend on how a torch build words its stream-mismatch warning."""
    wanted = {id(p) for p in params}
    found = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                            # isinstance() on lazy module attributes may warn
        for obj in gc.get_objects():
            try:
                if (isinstance(obj, torch.Tensor) and obj.grad_fn is not None and obj.device == device
                        and _reaches(obj.grad_fn, wanted)):
                    found.append(obj)
            except (ReferenceError, RuntimeError):                 # objects dying under the walk
                continue
    return found


class GraphedTrainStep:
    """One optimisation step of a ``WHVINetwork`` -- ``loss(x, y, n)``, ``backward()``, ``optimizer.step()`` and, when
    given, ``scheduler.step()`` -- recorded into a hipGraph and replayed: ``loss = step(x, y)``.

    The optimizer must keep its state on the device (``torch.optim.Adam(..., capturable=True)``).  A learning-rate
    schedule must live there too: the reference's recipe steps its ``LambdaLR`` after every batch (src/networks.py:80-81,
    src/evaluation.py:25-26), which a host-side scheduler cannot do inside a graph; ``make_optimizer(net,
    capturable=True)`` returns Adam with a device learning rate and a ``DeviceLambdaLR`` whose ``step()`` is captured
    with the rest.  ``x`` / ``y`` keep the example's shapes; the returned loss is a static tensor overwritten by the
    next call.  Fresh eps are drawn on every replay -- unless ``static_eps=True``: every WHVI layer then reads its draws
    from a static buffer (``step.eps_buffers``, one ``(J, S, D)`` tensor per layer in module order) that the caller
    fills before each call: how recorded trajectories are replayed through the captured step.

    Building it has NO side effects on the training state: the warm-up steps that torch needs before a capture
    (optimizer state initialised, allocator warm) run on a side stream, and parameters, optimizer state, schedule, the
    device's generator and the in-kernel Philox states are then put back exactly where they were.  At least one warm-up
    step always runs.

    It may be built after the network has trained eagerly, provided no tensor of an earlier pass that still carries
    an autograd graph (a kept ``loss``) is alive: gradient accumulators created by a backward pass on another
    stream would make the capture abort the PROCESS inside the HIP runtime.  That precondition is checked twice and a
    ``RuntimeError`` is raised BEFORE anything is touched: live tensors whose graph reaches the network's parameters
    are looked for directly, and the side-stream warm-up runs with torch's "AccumulateGrad node's stream does not
    match" warning promoted as a second line.  ``WHVINetwork.loss`` itself keeps only detached monitoring values."""

    def __init__(self, net, optimizer, example_x, example_y, n: int, ignore_kl: bool = False, warmup: int = 3,
                 scheduler=None, static_eps: bool = False):
        if example_x.device.type != "cuda":
            raise RuntimeError("GraphedTrainStep needs GPU tensors")
        for group in optimizer.param_groups:
            if "capturable" in group and not group["capturable"]:
                raise RuntimeError("GraphedTrainStep: create the optimizer with capturable=True")
        if scheduler is not None and not getattr(scheduler, "device_resident", False):
            raise RuntimeError("GraphedTrainStep: a scheduler that sets the learning rate from the host cannot be captured; "
                               "use whvi_amd.evaluation.DeviceLambdaLR (make_optimizer(net, capturable=True))")
        if scheduler is not None and not all(torch.is_tensor(g["lr"]) for g in optimizer.param_groups):
            raise RuntimeError("GraphedTrainStep: with a scheduler the optimizer's learning rate must be a device tensor")
        self.net, self.optimizer, self.scheduler = net, optimizer, scheduler
        self.n, self.ignore_kl = int(n), bool(ignore_kl)
        self.static_x, self.static_y = example_x.detach().clone(), example_y.detach().clone()
        dev = example_x.device
        params = [p for group in optimizer.param_groups for p in group["params"]]
        # gradient accumulators of an earlier eager backward belong to the stream that ran it; they die with the
        # last reference to that pass's autograd graph, so drop ours before looking for anybody else's
        net._pass_kl = None
        for module in net.modules():
            if hasattr(module, "_mc_kl"):
                module._mc_kl = None
        optimizer.zero_grad(set_to_none=True)
        gc.collect()
        stale_message = (
            "GraphedTrainStep: an autograd graph of an earlier pass is still alive (a kept `loss` tensor, a list of "
            "losses, ...): its gradient accumulators belong to another stream and capturing a backward pass "
            "through them would abort inside the HIP runtime. Drop those tensors (keep `loss.detach()` or "
            "`float(loss)` instead) and build the GraphedTrainStep again.")
        if _stale_graph_holders(params, dev):
            raise RuntimeError(stale_message)                # nothing has been touched yet
        torch.cuda.synchronize(dev)
        # ---- what the warm-up must not change
        saved_params = [p.detach().clone() for p in params]
        fresh_state = {id(p): (p not in optimizer.state or len(optimizer.state[p]) == 0) for p in params}
        saved_state = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in optimizer.state.get(p, {}).items()}
                       for p in params}
        saved_lrs = [g["lr"].clone() if torch.is_tensor(g["lr"]) else g["lr"] for g in optimizer.param_groups]
        saved_t = scheduler.t.clone() if scheduler is not None else None
        saved_rng = torch.cuda.get_rng_state(dev)
        philox = [(m, m._rng_state.clone()) for m in net.modules() if torch.is_tensor(getattr(m, "_rng_state", None))]
        if static_eps:
            for module in net.modules():
                if hasattr(type(module), "inkernel_rng") and not torch.is_tensor(getattr(module, "_eps_static", None)):
                    module._eps_static = True                # allocated by the layer's first draw (the warm-up); a buffer
                                                             # another captured step already reads is kept and shared
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        # A stale autograd graph keeps the parameters' AccumulateGrad nodes alive on the stream that created them; the
        # warm-up's backward on the side stream then has to synchronise with that stream -- harmless here, a process
        # abort inside torch.cuda.graph.  torch warns about exactly that (once per process unless warn-always is on).
        warn_always = torch.is_warn_always_enabled()
        torch.set_warn_always(True)
        try:
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                with torch.cuda.stream(side):
                    for _ in range(max(1, warmup)):
                        self._eager_step()
        finally:
            torch.set_warn_always(warn_always)
        torch.cuda.current_stream(dev).wait_stream(side)
        # ---- put the training state back where it was
        with torch.no_grad():
            for p, value in zip(params, saved_params):
                p.copy_(value)
            for p in params:
                state = optimizer.state.get(p, {})
                for key, value in state.items():
                    if not torch.is_tensor(value):
                        continue
                    if fresh_state[id(p)]:
                        value.zero_()                        # what Adam's lazy initialisation would have created
                    else:
                        value.copy_(saved_state[id(p)][key])
            for group, lr in zip(optimizer.param_groups, saved_lrs):
                if torch.is_tensor(lr):
                    group["lr"].copy_(lr)
            if scheduler is not None:
                scheduler.t.copy_(saved_t)
            for module, value in philox:
                module._rng_state.copy_(value)
        torch.cuda.set_rng_state(saved_rng, dev)
        self._eps_owners = [(m, m._eps_static) for m in net.modules() if torch.is_tensor(getattr(m, "_eps_static", None))]
        self.eps_buffers = [buf for _, buf in self._eps_owners]
        # the warm-up must leave no autograd graph -- hence no gradient accumulator of ITS stream -- behind: the capture's
        # forward would pick those accumulators up again (they live as long as any graph references them) and its backward,
        # on the capture stream, would then synchronise with the warm-up's stream inside the capture
        net._pass_kl = None
        for module in net.modules():
            if hasattr(module, "_mc_kl"):
                module._mc_kl = None
        optimizer.zero_grad(set_to_none=True)
        gc.collect()
        left_behind = _stale_graph_holders(params, dev)
        stale = [w for w in caught if "AccumulateGrad node's stream" in str(w.message)]
        for w in caught:                             # everything else is passed on unchanged
            if w not in stale:
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        if stale or left_behind:
            optimizer.zero_grad(set_to_none=True)
            raise RuntimeError(stale_message)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            loss = net.loss(self.static_x, self.static_y, self.n, ignore_kl=self.ignore_kl)
            loss.backward()
            optimizer.step()
            if scheduler is not None:
                scheduler.step()
            # keep the VALUE's static buffer, not the autograd graph recorded during the capture: that graph (and the
            # gradient accumulators it holds) must not outlive the capture, or a second captured step for the same
            # network -- another batch shape -- would find it as a stale graph of an earlier pass
            self.static_loss = loss.detach()
            del loss

    def _eager_step(self):
        self.optimizer.zero_grad(set_to_none=True)
        self.net.loss(self.static_x, self.static_y, self.n, ignore_kl=self.ignore_kl).backward()
        self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()

    def matches(self, x: torch.Tensor, y: torch.Tensor) -> bool:
        return x.shape == self.static_x.shape and y.shape == self.static_y.shape and x.dtype == self.static_x.dtype

    def release_static_eps(self):
        """Give the layers their generator back (the captured graph keeps reading the static buffers)."""
        for module, buf in self._eps_owners:
            if module._eps_static is buf:
                module._eps_static = None
        for module in self.net.modules():                    # containers that were marked but never drew themselves
            if getattr(module, "_eps_static", None) is True:
                module._eps_static = None

    def restore_static_eps(self):
        """Undo ``release_static_eps`` (eager steps and further captured steps beside this one read the same buffers)."""
        for module, buf in self._eps_owners:
            module._eps_static = buf

    def __call__(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        if x.shape != self.static_x.shape or y.shape != self.static_y.shape:
            raise RuntimeError("GraphedTrainStep: batch shape differs from the captured example")
        self.static_x.copy_(x)
        self.static_y.copy_(y)
        self.graph.replay()
        return self.static_loss
