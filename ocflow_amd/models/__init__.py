"""Host-side mirror of the reference's `models` package for the hot path: same module paths, class names, hparams,
method names and state_dict keys (reference: models/*.py), with the per-pair hot ops routed to ocflow_amd.ops."""
