"""File-level drivers of the auto-labeling run: `tools/static_eval.py` and `tools/dynamic_eval.py` of the reference
with every per-item Python step replaced by the batched device path of this package —

    trackStatic.pkl / trackDynamic.pkl + infos + annos/*.pkl + det_annos.pkl        (SURVEY.md 8(g) formats)
      -> crop preparation on the device            prep.prepare_*_batch            (N1; *TRACK.__getitem__)
      -> the heads + box decode                    model.refine                    (8(a); test_one_epoch)
      -> one all-gather of the boxes over RCCL when launched on several GPUs       (8(e))
      -> write-back into det_annos on the device   post.writeback_*                (N3; postprocessing)
      -> static/box/{model_type}.pkl / dynamic/box/box.pkl

Same command line as the reference's scripts (static_eval.py:291-299, dynamic_eval.py:247-254):

    python -m 3dal_pytorch_amd.eval static  --track trackStatic.pkl  --infos infos.pkl --model_path x.pth \
                                            --model_type one_box_est --det_annos det_annos.pkl
    python -m 3dal_pytorch_amd.eval dynamic --track trackDynamic.pkl --infos infos.pkl --model_path x.pth \
                                            --det_annos det_annos.pkl

Differences from the reference, all deliberate:
  * every annotation pickle is read once per run (the reference re-reads one per item, per frame and per
    window slot: static_model.py:536, static_eval.py:78,85, dynamic_model.py:449,464);
  * `--sampler numpy` (default) consumes the global NumPy stream in the reference's order — per batch the
    items' resampling draws, then the crops' object-point draws — so the same seed gives the same crops;
    `--sampler device` draws on the GPU from a counter-based generator keyed on the global item index, which is
    what a multi-GPU launch needs (a sharded run cannot share one host stream) and what throughput runs use;
  * a dynamic item whose own frame lacks the matched annotation: the reference's Dataset returns a RANDOM other
    item in its place (dynamic_model.py:487-489) and `postprocessing` then skips that frame
    (dynamic_eval.py:88-89), i.e. the substitute's box is computed and never read. With the numpy sampler this
    is reproduced draw for draw (the item's window draws, `randint`, the substitute's, recursively) so that every
    later item sees the reference's stream; with the device sampler the item itself is refined — either way the
    row is one the write-back does not use;
  * the IoU numbers `postprocessing` logs need the un-vendored fpointnet_train.provider_fpointnet
    (tools/utils.py:5,81-103) and are out of scope; the product of the run — the rewritten
    det_annos — is what this module writes.
"""
import argparse
import logging
import os
import pathlib
import pickle
import random

import numpy as np
import torch
import torch.distributed as dist

from . import dist as sharding
from . import post, prep

SEED = 10922081                                     # static_eval.py:303


def fix_seed(seed=SEED):
    """tools/utils.py:24-29"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def reorganize_info(infos):
    """list of info dicts -> {token: info} (tools/utils.py:46-51)"""
    return {info["token"]: info for info in infos}


def sort_detections(detections):
    """det_annos ordered by frame_id (static_eval.py:169-176)"""
    rank = np.argsort(np.array([det["frame_id"] for det in detections]))
    return [detections[r] for r in rank]


class Annos:
    """The per-frame annotation pickles ({split}/annos/*.pkl), each read at most once."""

    def __init__(self, infos):
        self.infos = infos
        self._cache = {}

    def __call__(self, token):
        a = self._cache.get(token)
        if a is None:
            with open(self.infos[token]["anno_path"], "rb") as f:
                a = pickle.load(f)
            self._cache[token] = a
        return a

    def pose(self, token):
        """flat-16 veh_to_global of the frame"""
        return np.asarray(self(token)["veh_to_global"], np.float64).reshape(16)

    def gt_box(self, token, name):
        """the (9,) `box` of the annotation called `name` in that frame, or None (the last object of that
        name, as the reference's loops without `break` leave it: static_eval.py:106-108; names are unique per frame)."""
        box = None
        for obj in self(token)["objects"]:
            if obj["name"] == name:
                box = obj["box"]
        return box


def token_to_det_index(infos, det_annos, annos):
    """{token: row of det_annos} through 'segment-{scene}_with_camera_labels_{frame:03d}' (static_eval.py:368-376)"""
    by_frame = {d["frame_id"]: i for i, d in enumerate(det_annos)}
    out = {}
    for token in infos:
        a = annos(token)
        out[token] = by_frame[f"segment-{a['scene_name']}_with_camera_labels_{a['frame_id']:03d}"]
    return out


def preprocessing(track, annos):
    """drop the static tracks whose best-score frame lacks the matched annotation (static_eval.py:26-44)"""
    for k in [k for k, v in track.items()
              if annos.gt_box(v["token"][int(np.argmax(np.stack(v["score"])))], v["match"][-1]) is None]:
        del track[k]
    return track


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _check_sampler(sampler, group):
    if sampler not in ("numpy", "device"):
        raise ValueError(f"unknown sampler {sampler!r}")
    if sampler == "numpy" and _world(group)[1] > 1:
        raise ValueError("sampler='numpy' draws from one host stream and cannot be sharded; use sampler='device'")


def refine_static_tracks(model, track, annos, batch_size=64, n_points=4096, sampler="numpy", seed=SEED, group=None):
    """`test_one_epoch` of static_eval.py:255-289 over STATICTRACK(track): (n_tracks, 7) float64 refined boxes,
    one per track, in the vehicle frame of the track's best-score frame. On several ranks each refines a
    contiguous range of tracks and the boxes are all-gathered once."""
    _check_sampler(sampler, group)
    tracks = list(track.values())
    n = len(tracks)
    dev = next(model.parameters()).device
    model.eval()
    model.sampler, model.seed = sampler, seed
    rank, world = _world(group)
    lo, hi = sharding.shard_range(n, rank, world)
    local = torch.zeros((hi - lo, 7), dtype=torch.float32, device=dev)
    for a in range(lo, hi, batch_size):
        b = min(a + batch_size, hi)
        chunk = tracks[a:b]
        poses = [annos.pose(t["token"][int(np.argmax(np.stack(t["score"])))]) for t in chunk]
        pts, init = prep.prepare_static_batch(chunk, poses, n_points=n_points, sampler=sampler, seed=seed,
                                              item_offset=a, device=dev)
        model.item_offset = a
        local[a - lo:b - lo] = model.refine(pts, init)
    model.item_offset = 0
    return sharding.all_gather_boxes(local, n, group).double().cpu().numpy()


def _dynamic_items(track, annos):
    """[(track_index, frame_index, has_gt)] in DYNAMICTRACK's index order (dynamic_model.py:406-424)"""
    items = []
    for t, v in enumerate(track.values()):
        for i, tok in enumerate(v["token"]):
            items.append((t, i, annos.gt_box(tok, v["match"][-1]) is not None))
    return items


def refine_dynamic_tracks(model, track, annos, batch_size=64, n_per_frame=1024, sampler="numpy", seed=SEED, group=None):
    """`test_one_epoch` of dynamic_eval.py:213-245 over DYNAMICTRACK(track): (sum of track lengths, 7) float64,
    one refined box per track-frame in that frame's vehicle frame (see the module docstring for the rows of
    frames without the matched annotation, which the write-back never reads)."""
    _check_sampler(sampler, group)
    tracks = list(track.values())
    items = _dynamic_items(track, annos)
    n = len(items)
    dev = next(model.parameters()).device
    model.eval()
    model.sampler, model.seed = sampler, seed
    rank, world = _world(group)
    lo, hi = sharding.shard_range(n, rank, world)
    local = torch.zeros((hi - lo, 7), dtype=torch.float32, device=dev)

    store = prep.TrackStore(tracks, dev)                    # every frame of every track, uploaded once

    def prepare(idx, first):
        """(pts, box, init) of items[idx]; the device generator is keyed on first + row"""
        return prep.prepare_dynamic_batch(store, [items[k][:2] for k in idx],
                                          [annos.pose(tracks[items[k][0]]["token"][items[k][1]]) for k in idx],
                                          n_per_frame=n_per_frame, r=model.r, s=model.s, sampler=sampler, seed=seed,
                                          item_offset=first, device=dev)

    def burn(k):
        t, it, _ = items[k]
        for i in range(it - model.r, it + model.r + 1):
            if 0 <= i < len(tracks[t]["point"]) and len(tracks[t]["point"][i]) > 0:
                np.random.choice(len(tracks[t]["point"][i]), n_per_frame, replace=True)

    for a in range(lo, hi, batch_size):
        b = min(a + batch_size, hi)
        if sampler == "device":
            pts, box, init = prepare(list(range(a, b)), a)
        else:
            parts, seg = [], []
            for k in range(a, b):
                while not items[k][2]:                      # replaced item: its draws, then the pick of its substitute
                    if seg:
                        parts.append(prepare(seg, 0))
                        seg = []
                    burn(k)
                    k = int(np.random.randint(n))
                seg.append(k)
            parts.append(prepare(seg, 0))
            if len(parts) == 1:
                pts, box, init = parts[0]
            else:                                           # joined in the point-major storage the views come from
                pts = torch.cat([p[0].transpose(2, 1) for p in parts], 0).transpose(2, 1)
                box = torch.cat([p[1].transpose(2, 1) for p in parts], 0).transpose(2, 1)
                init = torch.cat([p[2] for p in parts], 0)
        model.item_offset = a
        local[a - lo:b - lo] = model.refine(pts, box, init)
    model.item_offset = 0
    return sharding.all_gather_boxes(local, n, group).double().cpu().numpy()


def write_back(track, annos, token2idx, final_bboxes, det_annos, static):
    """the det_annos rewrite of `postprocessing` (static_eval.py:62-167, dynamic_eval.py:43-141): for every track
    frame that has the matched annotation, the first detection of that frame within 0.1 m of the track's box gets
    the refined box. det_annos is modified in place (like the reference) and returned."""
    tracks = list(track.values())
    tokens = sorted({t for v in tracks for t in v["token"]}, key=lambda t: token2idx[t])
    v2g = {t: annos.pose(t) for t in tokens}
    dets = {t: det_annos[token2idx[t]]["boxes_lidar"] for t in tokens}
    has_gt = {(i, t): annos.gt_box(t, v["match"][-1]) is not None for i, v in enumerate(tracks) for t in v["token"]}
    fn = post.writeback_static if static else post.writeback_dynamic
    new, _ = fn(tracks, v2g, has_gt, final_bboxes, dets)
    for t in tokens:
        rows = det_annos[token2idx[t]]["boxes_lidar"]
        rows[...] = new[t].astype(rows.dtype, copy=False)
    return det_annos


def _load(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def _logger(log_file):
    """console + file, as tools/utils.py:31-44; the handlers belong to one run() and are closed by it"""
    logger = logging.getLogger("3dal_pytorch_amd.eval")
    logger.setLevel(logging.INFO)
    logger.propagate = False
    fmt = logging.Formatter("%(asctime)s  %(levelname)5s  %(message)s")
    for h in (logging.StreamHandler(), logging.FileHandler(filename=log_file, mode="w")):
        h.setFormatter(fmt)
        logger.addHandler(h)
    return logger


def run(head, track_path, infos_path, det_annos_path, model_path, model_type="one_box_est", batch_size=64,
        sampler="numpy", device="cuda", result_path=None, group=None, precision="fp32"):
    """`main()` of static_eval.py:291-397 (head='static') / dynamic_eval.py:247-309 (head='dynamic').
    Returns (final_bboxes, det_annos); rank 0 writes the result pickle where the reference writes it."""
    from . import dynamic_model, static_model
    fix_seed(SEED)
    root = pathlib.Path(track_path).parent / head
    rank, _ = _world(group)
    result_path = pathlib.Path(result_path) if result_path else \
        root / "box" / (f"{model_type}.pkl" if head == "static" else "box.pkl")
    log_dir = root / "log" / "eval"
    if rank == 0:
        result_path.parent.mkdir(parents=True, exist_ok=True)
        log_dir.mkdir(parents=True, exist_ok=True)
    logger = _logger(log_dir / (f"{model_type}.txt" if head == "static" else "eval.txt")) if rank == 0 else None

    def say(msg):
        if logger:
            logger.info(msg)
    try:
        say("Load track data")
        track = _load(track_path)
        say("Load info data")
        infos = reorganize_info(_load(infos_path))
        det_annos = sort_detections(_load(det_annos_path))
        annos = Annos(infos)
        token2idx = token_to_det_index(infos, det_annos, annos)
        if head == "static":
            ctor = {"one_box_est": static_model.StaticModelOneBoxEst, "two_box_est": static_model.StaticModelTwoBoxEst}
            if model_type not in ctor:
                raise ValueError(f'No model supports for model type "{model_type}".')
            track = preprocessing(track, annos)
            model = ctor[model_type](n_classes=3, n_channel=3)
        elif head == "dynamic":
            model = dynamic_model.DynamicModel(n_classes=3, n_channel=4)
        else:
            raise ValueError(f"unknown head {head!r}")
        say(f"Load model from {model_path}")
        model.load_state_dict(torch.load(model_path, map_location="cpu")["model_state_dict"])
        model = model.to(device)
        model.precision = precision                 # "bf16"/"fp16": 16-bit MFMA operands, fp32 accumulate (C3/C5)
        say("Start testing")
        refine = refine_static_tracks if head == "static" else refine_dynamic_tracks
        final_bboxes = refine(model, track, annos, batch_size=batch_size, sampler=sampler, group=group)
        say("Start post processing")
        det_annos = write_back(track, annos, token2idx, final_bboxes, det_annos, static=(head == "static"))
        if rank == 0:
            say(f"Saving results to {result_path}")
            with open(result_path, "wb") as f:
                pickle.dump(det_annos, f)
        return final_bboxes, det_annos
    finally:
        if logger:
            for h in list(logger.handlers):
                logger.removeHandler(h)
                h.close()


def main(argv=None):
    parser = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    parser.add_argument("head", choices=["static", "dynamic"])
    parser.add_argument("--track", required=True, help="Path to trackStatic.pkl / trackDynamic.pkl.")
    parser.add_argument("--infos", required=True, help="Path to infos file.")
    parser.add_argument("--model_path", required=True, help="Path to model.")
    parser.add_argument("--model_type", default="one_box_est", help="Type of model (static head).")
    parser.add_argument("--det_annos", required=True, help="Path to detection annos.")
    parser.add_argument("--batch_size", type=int, default=64)
    parser.add_argument("--sampler", choices=["numpy", "device"], default="numpy")
    parser.add_argument("--precision", choices=["fp32", "f16x3", "bf16", "fp16"], default="fp32",
                        help="MFMA operand type of the shared MLPs (fp32 = the reference's arithmetic).")
    parser.add_argument("--result", default=None, help="Output pickle (default: the reference's location).")
    args = parser.parse_args(argv)
    group = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:                      # launched by torch.distributed.run
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    try:
        run(args.head, args.track, args.infos, args.det_annos, args.model_path, args.model_type, args.batch_size,
            args.sampler, result_path=args.result, group=group, precision=args.precision)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
