"""ocflow_amd -- MI355X-native hot path of OCFlow (cost volume, backward warp, photometric / smoothness /
range-map losses) as hand-written HIP kernels behind the reference's own operator and nn.Module surface.

    ocflow_amd.ops                     functional operators (torch.autograd.Function over the C ABI)
    ocflow_amd.models.flow_model       FlowModel        (reference: models/flow_model.py)
    ocflow_amd.models.flow_occ_model   FlowOccModel     (reference: models/flow_occ_model.py)
    ocflow_amd.models.model            FlowStageModel + loss functions (reference: models/model.py)
"""
__version__ = "0.1.0"

import os as _os

# ROCm 7.x hipGraph replays take a "packet capture" fast path (pre-baked AQL packets) by default.  On the ~2000-node graph of
# one OCFlow training step that path replays with stale zero-fills (reduction semaphores / split-K workspaces keep their
# previous contents) as soon as the host synchronises between two replays: gradients fill with garbage within 2-3 replays
# (scripts/long_run.py --every 1 reproduces it; DESIGN.md "hipGraph").  The regular enqueue path is correct and just as fast
# (measured 421 vs 426 pairs/s).  The flag is read when the HIP runtime initialises, so it is set at import, before any HIP
# call; runtime.TrainStepper.enable_graph refuses to capture when it is not in effect.  "In effect" is checked, not assumed: when the
# variable was absent and the process had already initialised HIP (torch.cuda in use before this import), setting it now changes
# nothing -- GRAPH_ENV_LATE records that, and enable_graph refuses.
import sys as _sys


def _hip_runtime_started() -> bool:
    """has THIS process already initialised the ROCm runtime?  torch's own flag (`torch.cuda.is_initialized()`) only covers torch's lazy
    initialisation: `torch.cuda.is_available()` / a ctypes call into libamdhip64 start the runtime -- which is when the variable is
    read -- without setting it.  The runtime opens /dev/kfd when it starts: an open descriptor on it is the process-level evidence."""
    try:
        for fd in _os.listdir("/proc/self/fd"):
            try:
                if _os.readlink(f"/proc/self/fd/{fd}") == "/dev/kfd":
                    return True
            except OSError:
                continue
    except OSError:
        pass
    return False


def _graph_env_guard() -> bool:
    """Set DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 unless the caller chose a value; True when that came too late to take effect."""
    if "DEBUG_CLR_GRAPH_PACKET_CAPTURE" in _os.environ:
        return False  # the launcher's choice (a value other than "0" is refused by enable_graph itself)
    torch = _sys.modules.get("torch")
    late = bool(torch is not None and torch.cuda.is_initialized()) or _hip_runtime_started()
    _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    return late


GRAPH_ENV_LATE = _graph_env_guard()
