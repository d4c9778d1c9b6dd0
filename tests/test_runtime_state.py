"""CPU: the gradient-sink registry of ocflow_amd.conv honours an entry only for the very tensor it was installed for (weak reference +
shape), so a stale entry or a re-used id() cannot route a gradient into another model's buffer."""
import torch

from ocflow_amd import conv as nconv
from ocflow_amd import ops


def test_sinks_are_keyed_by_identity_and_scoped(monkeypatch):
    monkeypatch.setattr(ops, "_take_zeros", lambda n, device: torch.zeros(n))
    p, q = torch.nn.Parameter(torch.randn(4, 3, 3, 3)), torch.nn.Parameter(torch.randn(4, 3, 3, 3))
    view = torch.zeros(4, 3, 3, 3)
    owner = object.__new__(type("Owner", (), {}))
    nconv.record_native(True)
    t, is_sink = nconv._grad_target(p, p.device)
    nconv.record_native(False)
    assert not is_sink and nconv._seen(p) and not nconv._seen(q)
    with nconv.grad_sinks([p, q], [view, torch.zeros(4, 3, 3, 3)], owner=owner) as ids:
        assert ids == {id(p)}                                   # only the parameter the native backward served
        assert nconv._grad_target(p, p.device)[0] is view
        # an entry under q's id that was installed for ANOTHER tensor (what a re-used id looks like) is not honoured, and is dropped
        nconv.GRAD_SINK[id(q)] = nconv.GRAD_SINK[id(p)]
        t, is_sink = nconv._grad_target(q, q.device)
        assert not is_sink and t is not view and id(q) not in nconv.GRAD_SINK
        # another owner's clear does not touch these sinks
        nconv.clear_grad_sinks(owner=object())
        assert nconv.GRAD_SINK
    assert not nconv.GRAD_SINK
    # a block that raises clears them too
    try:
        with nconv.grad_sinks([p], [view], owner=owner):
            assert nconv.GRAD_SINK
            raise KeyError("x")
    except KeyError:
        pass
    assert not nconv.GRAD_SINK
    # a sink of the wrong shape is never installed
    assert nconv.install_grad_sinks([p], [torch.zeros(4, 3, 3)]) == set()
    nconv.clear_grad_sinks()


def test_retired_descriptor_tables_follow_their_owners():
    pk = nconv.WeightPacker()

    class Owner:
        pass

    a, b = Owner(), Owner()
    pk._dev_descs = torch.zeros(4)
    pk._retire_table()                       # never captured: dropped at once
    assert pk._retired == [] and pk._dev_descs is None
    pk._dev_descs = torch.zeros(4)
    with pk.capture_owner(a):
        pk._note_capture()
    with pk.capture_owner(b):
        pk._note_capture()
    pk._retire_table()
    assert len(pk._retired) == 1
    pk.release_owner(a)
    assert len(pk._retired) == 1             # b's graph may still replay it
    del b
    import gc
    gc.collect()
    pk._sweep_retired()
    assert pk._retired == []
    pk._dev_descs = torch.zeros(4)
    pk._note_capture()                       # captured outside any owner scope: kept for the life of the process
    pk._retire_table()
    pk.release_owner(a)
    assert len(pk._retired) == 1 and pk._retired[0][1] is None
