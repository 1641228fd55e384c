"""CLI -- ``python -m clane_amd [embedding] --data_root D --output_root O --config_file C``.

Mirrors the reference's ``clane/__main__.py`` (same flags, same YAML keys, same outputs:
``output_root/Z.npy`` and, with ``--save_history``, ``output_root/{outer}/Z_{sweep}.npy``).
The README form ``clane embedding ...`` (README.md:11) is accepted too: the reference's parser
rejects the ``embedding`` token, here it is an optional no-op.  The loop always
runs on the GPU; ``--gpu`` only selects where ``Embedder.device`` points, as upstream.
"""
from __future__ import annotations

import argparse
import os
import shutil
from pathlib import Path

import numpy as np
import torch
import yaml

from . import similarity
from .embedder import Embedder, IterativeEmbedder
from .graph import Graph


def _distributed_setup():
    """Under `torchrun --nproc-per-node N` (WORLD_SIZE > 1): one process per GPU, RCCL process group.
    Returns (rank, world).  A plain `python -m clane_amd` run is (0, 1) and touches nothing.
    Rehearsal on a one-GPU box: CLANE_DIST_BACKEND=gloo CLANE_SHARE_GPU=1 (RCCL refuses two ranks on one device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1
    import torch.distributed as dist
    local_rank = 0 if os.environ.get("CLANE_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.device_count() == 1:              # the launcher masked the devices per process
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        if os.environ.get("CLANE_DIST_BACKEND", "nccl") == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    return dist.get_rank(), world


def embedding(args):
    rank, world = _distributed_setup()
    say = print if rank == 0 else (lambda *a, **k: None)      # every rank computes; rank 0 talks and writes
    say('[Embedding]', end='\n')

    if args.config_file.absolute().exists():
        with open(args.config_file.absolute(), 'r') as config_io:
            hparams = yaml.load(config_io, Loader=yaml.FullLoader)
    else:
        raise FileNotFoundError(f"Config file not found. {args.config_file.absolute()}")

    device = torch.device('cuda') if args.gpu else torch.device('cpu')
    g = Graph(data_root=args.data_root, **hparams["graph"])

    if world > 1:                                   # unseeded N(0,1) content (no C.npy) must agree across ranks
        import torch.distributed as dist
        Xd = g.X.cuda()
        dist.broadcast(Xd, 0)
        g.X = Xd.cpu()

    say("Graph Loaded.")
    say(f" - {len(g)} vertices")
    say(f" - {len(g.E)} edges")
    say(" - Content Embeddings:")
    say(f"     - dim : {g.d:3d}")
    say(f"     - mean: {g.X.mean():5.2f}")
    say(f"     - std : {g.X.std():5.2f}")

    try:
        similarity_measure = getattr(similarity, hparams["similarity"]["method"])
    except AttributeError:
        raise AttributeError(f'Given similarity method {hparams["similarity"]["method"]} not found.')

    similarity_measure = similarity_measure(**hparams['similarity']['kwargs'])

    embedder_cls = IterativeEmbedder if hasattr(similarity_measure, 'parameters') else Embedder
    extra = {"num_workers": args.num_workers} if embedder_cls is IterativeEmbedder else {}
    if world > 1 and getattr(args, "exchange", "auto") != "auto":
        g.engine(device if device.type == "cuda" else None, exchange=args.exchange)     # first use fixes the division
    if args.init_Z is not None:                     # resume: start from saved embeddings instead of Z = X
        Z0 = torch.from_numpy(np.load(args.init_Z))
        if tuple(Z0.shape) != tuple(g.X.shape):
            raise ValueError(f"--init_Z holds shape {tuple(Z0.shape)}, the graph needs {tuple(g.X.shape)}")
        g.set_Z(Z0.to(g.X.dtype))
    if args.save_history and embedder_cls is Embedder:
        # write output_root/{outer}/Z_{sweep}.npy (the reference's layout, __main__.py:73-86) WHILE the sweeps run
        def write_sweep(outer, sweep, Z):
            if rank == 0:
                folder = args.output_root.joinpath(f'{outer}')
                folder.mkdir(parents=True, exist_ok=True)
                np.save(folder.joinpath(f'Z_{sweep}.npy'), _to_numpy(Z))
        extra["history_sink"] = write_sweep
        if world > 1:       # every rank stages its own part of Z; rank 0's writer thread assembles them (one box: same disk)
            extra["history_parts_dir"] = args.output_root.joinpath(".clane_history_parts")
    embedder = embedder_cls(graph=g, similarity_measure=similarity_measure, device=device,
                            save_history=args.save_history, **extra, **hparams["embedder"])
    if rank != 0:
        embedder.verbose = False
    embedder.iterate()
    final_Z = g.Z                                   # collective when world > 1: every rank takes part
    if "history_parts_dir" in extra and rank == 0:
        shutil.rmtree(extra["history_parts_dir"], ignore_errors=True)      # iterate() flushed: every part was consumed
    if world > 1:                                   # leave together: nobody tears the group down while others talk
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    say("Saving the results.")
    if not args.output_root.exists():
        args.output_root.mkdir(parents=True, exist_ok=True)

    np.save(args.output_root.joinpath('Z.npy'), _to_numpy(final_Z))

    say(f"The embeddings are stored in {args.output_root.joinpath('Z.npy').absolute()}.")


def _to_numpy(Z: torch.Tensor) -> np.ndarray:
    Z = Z.cpu()
    return (Z.float() if Z.dtype == torch.bfloat16 else Z).numpy()      # NumPy has no bfloat16


def get_parser():
    parser = argparse.ArgumentParser(prog="clane")
    parser.add_argument("command", nargs="?", choices=["embedding"], default="embedding",
                        help="Optional; the only command is 'embedding'.")
    parser.add_argument("--data_root", type=Path, help="Path to the data root directory.")
    parser.add_argument("--output_root", type=Path, help="Path to the root for the experiment results to be stored.")
    parser.add_argument("--config_file", type=Path, help="Path to the training configuration yaml file.")
    parser.add_argument("--save_history", action='store_true',
                        help="If true, it saves the embeddings for every iteration.")
    parser.add_argument("--num_workers", type=int, default=0)
    parser.add_argument("--init_Z", type=Path, default=None,
                        help="(extension) .npy of shape [V, d]: start from these embeddings instead of the content "
                             "embeddings, e.g. the Z.npy of an interrupted run.")
    parser.add_argument("--exchange", default="auto",
                        choices=["auto", "columns", "allgather_all", "allgather", "halo", "halo_p2p"],
                        help="(extension, multi-GPU runs under torchrun) how the sweep is divided over the GPUs: columns of "
                             "Z (no exchange per sweep; what auto picks for wide rows), rows with one all-gather of the "
                             "updated rows per sweep (allgather_all) or the leaner row splits; "
                             "see DESIGN.md section 6.  A plug-in similarity needs a row division (default then: halo).")
    parser.add_argument("--gpu", action='store_true')
    return parser


def main():
    parser = get_parser()
    args = parser.parse_args()
    embedding(args)


if __name__ == "__main__":
    main()
