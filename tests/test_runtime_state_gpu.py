"""Process-global state of the training runtime has an owner and does not outlive it: gradient
sinks exist only inside the capture that installed them -- also when that capture raises --, the packed convolution weights follow the
captured optimizer, the packer's descriptor tables die with the graphs that could replay them."""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _stepper(model="flowoccnet", amp=torch.bfloat16, lr=1e-4):
    from ocflow_amd.models.flow_occ_model import FlowOccModel
    from ocflow_amd.runtime import DistEnv, TrainStepper, synthetic_batch
    torch.manual_seed(5)
    env = DistEnv(device=torch.device(DEV))
    st = TrainStepper(FlowOccModel({"model": model, "learning_rate": lr}), env, amp_dtype=amp, channels_last=True, fused_optimizer=True,
                      master_weights=True)
    batch = st.prepare(synthetic_batch(2, 128, 192, env.device, seed=3))
    return st, batch


def test_a_capture_that_raises_leaves_no_sinks_and_the_next_stepper_trains(monkeypatch):
    from ocflow_amd import conv as nconv
    st, batch = _stepper()
    st.step(batch)
    real = st._gather

    def boom(*a, **k):
        raise RuntimeError("injected failure inside the capture")

    monkeypatch.setattr(st, "_gather", boom)
    with pytest.raises(RuntimeError, match="injected"):
        st.enable_graph(batch)
    assert not nconv.GRAD_SINK and not nconv._RECORD[0] and st._graph is None and st._flat is None
    monkeypatch.setattr(st, "_gather", real)
    # the same process goes on: eager steps of the failed stepper, then a NEW stepper (whose parameters may re-use the ids) with a graph
    l0 = float(st.step(batch))
    st.close()
    del st
    gc.collect()
    st2, batch2 = _stepper()
    st2.step(batch2)
    st2.enable_graph(batch2)
    assert not nconv.GRAD_SINK  # sinks live inside the capture only
    losses = [float(st2.step(st2._static_batch)) for _ in range(6)]
    assert all(l == l and abs(l) < 1e4 for l in losses + [l0]), (l0, losses)
    st2.close()


def test_packed_weights_follow_the_captured_optimizer():
    """k replays, then an eager no-grad forward: the implicit-GEMM layers must read the weights of step k, not k-1 (the captured Adam
    moves no version counter and runs no optimizer hook)"""
    from ocflow_amd import conv as nconv
    st, batch = _stepper()
    st.step(batch)
    st.enable_graph(batch)
    for _ in range(4):
        st.step(st._static_batch)
    st.raw.eval()
    x = st._static_batch[0]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        got = [t.float().clone() for t in st.raw(x)]
        nconv.weights_changed()
        nconv.PACKER.refresh()  # explicit re-pack from the live weights
        want = [t.float().clone() for t in st.raw(x)]
    for g, w in zip(got, want):
        assert torch.equal(g, w), float((g - w).abs().max())
    st.close()


def test_descriptor_tables_die_with_their_graphs():
    from ocflow_amd import conv as nconv
    st, batch = _stepper()
    st.step(batch)
    st.enable_graph(batch)
    st.step(st._static_batch)
    torch.cuda.synchronize()
    owners_before = nconv.PACKER._table_owners
    assert owners_before and any(r() is st for r in owners_before)  # the capture registered its stepper on the table it replays
    # a second model registers new weights: the first stepper's table is retired, but kept -- its graph still replays it
    st_b, batch_b = _stepper()
    st_b.step(batch_b)
    kept = [t for t, o in nconv.PACKER._retired if o is not None and any(r() is st for r in o)]
    assert kept
    st.step(st._static_batch)  # (and it still replays correctly)
    torch.cuda.synchronize()
    st.close()
    assert not [t for t, o in nconv.PACKER._retired if o is not None and any(r() is st for r in o)]
    n_before = len(nconv.PACKER._retired)
    st_b.close()
    del st, st_b
    gc.collect()
    nconv.PACKER._purge()
    # the weights of both models are gone: so are their entries (what other tests left alive stays)
    assert len(nconv.PACKER._retired) <= n_before and all(e.src_ref() is not None for e in nconv.PACKER._order)


def test_side_stream_weight_gradients_match_the_single_stream_step(monkeypatch):
    """weight gradients launched beside the input gradients (conv.wgrad_bias_async: a fork / join inside the captured graph) against the
    same capture on one stream.  Learning rate 0: the weights stay where the seed put them (Adam's first steps are sign-like, so with a
    real learning rate the order of the fp32 atomics decides the update of every near-zero gradient and the two runs drift apart), every
    replay computes the same gradients up to that order"""
    from ocflow_amd import conv as nconv
    flats, losses = [], []
    for mode in ("0", "1"):
        monkeypatch.setenv("OCFLOW_WGRAD_STREAM", mode)
        st, batch = _stepper(lr=0.0)
        st.step(batch)
        st.enable_graph(batch)
        assert not any(nconv._WGRAD_PENDING.values())  # joined before the capture ended
        ls = [float(st.step(st._static_batch)) for _ in range(3)]
        torch.cuda.synchronize()
        flats.append(st._flat.clone())
        losses.append(ls)
        st.close()
        del st
        gc.collect()
    assert nconv._WGRAD_SIDE  # the second run did fork
    g0, g1 = flats
    assert torch.isfinite(g1).all() and float(g0.norm()) > 0
    rel = float((g0 - g1).norm() / g0.norm())
    assert rel < 2e-2, rel
    assert all(abs(a - b) <= 1e-3 * max(1.0, abs(a)) for a, b in zip(*losses)), losses
