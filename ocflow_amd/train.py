"""`python -m ocflow_amd.train config.yml` -- the reference's training launcher without Lightning.

Reproduces test_unsupervised.py:35-128 for `network_type: flow` (and test.py:17-84 for the supervised surfaces): seed 42, read
the YAML into `args`, build `hparams` exactly as the reference does (:44, :52-57), pick the module, fit.  The reference's
`hparams['using_pretrained_inpainting']` KeyError for `network_type: flow` (:124) is read with `.get(..., False)`.
`Trainer.fit` is replaced by a plain loop over `TrainStepper` (bf16 autocast, channels_last, optional hipGraph replay, one
process per GPU under torch.distributed.run).  Data: with `root` pointing at an MPI-Sintel tree, `ocflow_amd.data.datasets.DatasetModule`
(the reference's lightning_datamodule.py: `dataset_name`, 80/10/10 split with seed 42) in its decode-only mode -- crop, resize,
normalisation, flow rescaling and occlusion binarisation run batched on the training GPU (`prepare_batch_on_device`); with no dataset on
the box (`root` missing), synthetic Sintel-shaped batches of `image_size`.

    python -m ocflow_amd.train config/unsupervised_config.yml --steps 200 --override network_type=flow model=pwc
"""
from __future__ import annotations

import argparse
import json
import os
import time

import torch
import yaml

SUPERVISED = {"flow": "FlowModel", "flowocc": "FlowOccModel"}


def load_config(path: str) -> dict:
    with open(path) as f:
        return yaml.load(f, Loader=yaml.FullLoader)  # the reference's configs use !!python/tuple (test_unsupervised.py:38)


def build_module(args: dict):
    """(module, hparams) as test_unsupervised.py:44-58 / test.py:36-50 build them"""
    network_type = args["network_type"]
    if network_type == "flow" and "photo_weight" in args:  # unsupervised stage (test_unsupervised.py:49-58)
        from ocflow_amd.models.model import FlowStageModel
        hparams = dict(network_type=network_type, model=args["model"], epochs=args["epochs"], batch_size=args["batch_size"],
                       learning_rate=args["learning_rate"], log_every_n_steps=args.get("log_every_n_steps", 20),
                       img_size=args.get("image_size", None), org=args.get("org", False))
        assert hparams["model"] in ["simple", "flownets", "flownetc", "pwc", "flownet", "eflownet", "eflownet2"]
        for k in ("photo_weight", "smooth1_weight", "smooth2_weight", "with_occ", "occ_aware", "displacement"):
            hparams[k] = args[k]
        return FlowStageModel(hparams), hparams
    hparams = dict(args)
    if network_type == "flow":  # supervised (test.py:38-42)
        from ocflow_amd.models.flow_model import FlowModel
        return FlowModel(hparams), hparams
    if network_type == "flowocc":  # test.py:46-48
        from ocflow_amd.models.flow_occ_model import FlowOccModel
        return FlowOccModel(hparams), hparams
    if network_type == "twostage":  # test_unsupervised.py:72-98
        from ocflow_amd.models.model import TwoStageModel, TwoStageModelGC
        assert args["model"] in ["with_gt_flow", "no_gt_flow"]
        hparams = dict(network_type=network_type, model=args["model"], epochs=args["epochs"], batch_size=args["batch_size"],
                       learning_rate=args["learning_rate"], log_every_n_steps=args.get("log_every_n_steps", 20),
                       img_size=args.get("image_size", None), org=args.get("org", False))
        for k in ("reconst_weight", "photo_weight", "inpainting_root", "smooth1_weight", "smooth2_weight", "using_pretrained_inpainting",
                  "loss_type", "pixelwise_weight"):
            hparams[k] = args[k]
        if hparams["model"] == "no_gt_flow":
            hparams["smoothness_weight"] = args["smoothness_weight"]
            hparams["flow_root"] = args["flow_root"]
            hparams["supervised_flow"] = bool(args["supervised_flow"])
            return TwoStageModel(hparams), hparams
        hparams["inpainting_stage"] = args["inpainting_stage"]
        assert hparams["inpainting_stage"] in ["gated", "gated_org", "simple"]
        return TwoStageModelGC(hparams), hparams
    if network_type == "inpainting" and args.get("model", "simple") == "simple" and "loss_type" not in args:
        from ocflow_amd.models.inpainting_model import InpaintingModel
        return InpaintingModel(hparams), hparams
    raise NotImplementedError(f"network_type '{network_type}' is outside the hot path of the MI355X build (SURVEY.md sections 2.1, 8f-4)")


def main(argv=None):
    p = argparse.ArgumentParser(prog="python -m ocflow_amd.train")
    p.add_argument("config")
    p.add_argument("--steps", type=int, default=None, help="stop after this many steps (default: epochs x steps_per_epoch)")
    p.add_argument("--steps-per-epoch", type=int, default=100)
    p.add_argument("--override", nargs="*", default=[], help="key=value pairs applied to the YAML (values parsed as YAML)")
    p.add_argument("--graph", type=int, default=1, help="replay the step as a hipGraph (fixed-shape batches)")
    p.add_argument("--fp32", action="store_true")
    p.add_argument("--out", default=None, help="directory for the final state_dict + metrics.json")
    a = p.parse_args(argv)

    from ocflow_amd.runtime import TrainStepper, init_distributed, shutdown_distributed, synthetic_batch, use_repo_miopen_db
    use_repo_miopen_db()
    torch.manual_seed(42)  # pl.seed_everything(42), test_unsupervised.py:35
    args = load_config(a.config)
    for kv in a.override:
        k, v = kv.split("=", 1)
        args[k] = yaml.safe_load(v)
    # (test_unsupervised.py:124 reads hparams['using_pretrained_inpainting'] for every network type: KeyError for 'flow', SURVEY F7;
    # here only the two-stage branch looks at it, and TwoStageModelGC refuses a pretrained completion network by name)
    module, hparams = build_module(args)
    env = init_distributed()
    if env.device.type != "cuda":
        raise SystemExit("ocflow_amd.train needs an MI355X: the hot path has no CPU fallback")
    amp = None if a.fp32 else torch.bfloat16
    stepper = TrainStepper(module, env, amp_dtype=amp, channels_last=True, fused_optimizer=True,
                           dp_mode="flat" if (a.graph and env.world_size > 1) else "ddp", master_weights=amp is not None)
    B = int(args["batch_size"])
    per_rank = max(1, B // env.world_size)
    size = args.get("image_size") or (436, 1024)
    loader = None
    if args.get("root") and os.path.isdir(str(args["root"])):
        # test_unsupervised.py:105-107: DatasetModule(root, image_size, batch_size, dataset_name, overfit) -> prepare_data(); setup()
        from ocflow_amd.data.datasets import DatasetModule
        dm = DatasetModule(root=str(args["root"]), image_size=args.get("image_size"), batch_size=per_rank, dataset_name=args.get("dataset_name", "MpiSintelClean"),
                           num_workers=int(args.get("num_workers", 6)), overfit=bool(args.get("overfit", args.get("overfit_batches", False))), raw=True)
        dm.prepare_data()
        dm.setup()

        def batches():
            epoch = 0
            while True:
                n = 0
                for bi, raw in enumerate(dm.train_dataloader()):
                    if raw[0].shape[0] != per_rank or bi % env.world_size != env.rank:
                        continue  # (a fixed-shape graph replays full batches only; ranks take every world_size-th batch of one shuffle)
                    b = dm.to_device(raw, env.device)
                    if len(b) == 1:  # image-only datasets: the unsupervised losses need no labels, the step logs against zeros
                        b = (b[0], torch.zeros((per_rank, 2) + tuple(b[0].shape[2:]), device=env.device))
                    n += 1
                    yield stepper.prepare(b)
                epoch += 1
                if n == 0:
                    raise SystemExit(f"dataset under {args['root']} yields no full batch of {per_rank}")

        loader = batches()
    total = a.steps if a.steps is not None else int(args.get("epochs", 1)) * a.steps_per_epoch
    if loader is not None:
        batch = next(loader)
    else:
        batch = stepper.prepare(synthetic_batch(per_rank, int(size[0]), int(size[1]), env.device, seed=42, index_offset=env.rank * per_rank))
    loss = stepper.step(batch, 0)
    if a.graph:
        stepper.enable_graph(batch)
        batch = stepper._static_batch
    log_every = int(hparams.get("log_every_n_steps", 20))
    t0, history = time.perf_counter(), []
    for i in range(1, total):
        loss = stepper.step(next(loader) if loader is not None else batch, i)
        if i % log_every == 0 or i == total - 1:
            lv = float(loss)
            history.append({"step": i, "train_loss": lv})
            if env.is_main:
                dt = time.perf_counter() - t0
                print(f"step {i}: train_loss {lv:.5f}  ({per_rank * env.world_size * i / dt:.1f} pairs/s)", flush=True)
            if lv != lv:
                raise SystemExit(f"non-finite loss at step {i}")
    torch.cuda.synchronize()
    if a.out and env.is_main:
        os.makedirs(a.out, exist_ok=True)
        torch.save(stepper.module_state_dict(), os.path.join(a.out, "state_dict.pt"))  # same keys as the reference's save_state_dict
        with open(os.path.join(a.out, "metrics.json"), "w") as f:
            json.dump({"hparams": {k: (list(v) if isinstance(v, tuple) else v) for k, v in hparams.items()}, "history": history}, f)
    shutdown_distributed()
    return history


if __name__ == "__main__":
    main()
