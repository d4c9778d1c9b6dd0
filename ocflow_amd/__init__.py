"""ocflow_amd -- MI355X-native hot path of OCFlow (cost volume, backward warp, photometric / smoothness /
range-map losses) as hand-written HIP kernels behind the reference's own operator and nn.Module surface.

    ocflow_amd.ops                     functional operators (torch.autograd.Function over the C ABI)
    ocflow_amd.models.flow_model       FlowModel        (reference: models/flow_model.py)
    ocflow_amd.models.flow_occ_model   FlowOccModel     (reference: models/flow_occ_model.py)
    ocflow_amd.models.model            FlowStageModel + loss functions (reference: models/model.py)
"""
__version__ = "0.1.0"
