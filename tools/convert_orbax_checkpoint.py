"""Offline Orbax -> safetensors converter for LAP checkpoints.

    python tools/convert_orbax_checkpoint.py <orbax params dir> <out dir> [--config lap_libero] [--dtype float32]

Run it ONCE where `orbax-checkpoint` (and its jax / numpy stack) is installed — e.g. the environment the checkpoint was
trained or downloaded in; the MI355X image has no Orbax.  Input: the `params` item of a reference
checkpoint (`checkpoints/<config>/<exp>/<step>/params`, or a released `lihzha/LAP-3B[-Libero]` `params` directory),
restored the way the reference does it (src/lap/training/weight_loaders.py:143-189: PyTreeCheckpointer, numpy leaves,
the trailing "value" key of nnx.State stripped).  Output: `<out dir>/params.safetensors`, the same tree flattened with
'/' and prefixed `params/` — exactly what `lap_amd.checkpoints.restore_params` reads, so

    TrainConfig(weight_loader=WeightLoaderChoice(kind="checkpoint", params_path="<out dir>"))      # fine-tuning
    lap_amd.serve.create_trained_policy(config, "<checkpoint dir holding params/ and assets/>")     # serving

load it.  With `--config` the converted tree is validated against that model's parameter tree (unexpected / missing
keys, shapes) before it is written.  `flatten_tree` / `write_params` hold no Orbax dependency and are unit-tested here
on hand-built trees; only `restore_orbax` needs the package.
"""
from __future__ import annotations

import argparse
import os
import pathlib
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def restore_orbax(params_dir: str) -> dict:
    """The reference's restore_params (weight_loaders.py:143-189) with numpy leaves."""
    import jax
    import orbax.checkpoint as ocp

    path = pathlib.Path(params_dir).resolve()
    with ocp.PyTreeCheckpointer() as ckptr:
        meta = ckptr.metadata(path)
        tree = getattr(meta, "item_metadata", meta)
        tree = getattr(tree, "tree", tree)
        args = jax.tree.map(lambda _: ocp.ArrayRestoreArgs(restore_type=np.ndarray), tree)
        return ckptr.restore(path, ocp.args.PyTreeRestore(item=tree, restore_args=args))


def flatten_tree(tree: dict, sep: str = "/") -> dict:
    """Nested dict -> {'a/b/c': leaf}; the trailing 'value' key nnx.State adds to every leaf is removed when ALL leaves
    carry it (weight_loaders.py:184-189), and a single top-level 'params' wrapper is unwrapped."""
    flat = {}

    def walk(node, prefix):
        if isinstance(node, dict):
            for k, v in node.items():
                walk(v, prefix + (str(k),))
        else:
            flat[prefix] = node

    walk(tree, ())
    if flat and all(k[-1] == "value" for k in flat):
        flat = {k[:-1]: v for k, v in flat.items()}
    if flat and all(k[0] == "params" for k in flat):
        flat = {k[1:]: v for k, v in flat.items()}
    return {sep.join(k): v for k, v in flat.items()}


def write_params(flat: dict, out_dir, *, config: str | None = None, dtype: str = "float32", allow_partial: bool = False) -> pathlib.Path:
    import torch
    from safetensors.torch import save_file

    tensors = {}
    for k, v in flat.items():
        a = np.asarray(v)
        if a.dtype.kind == "V" or str(a.dtype) == "bfloat16":       # ml_dtypes bfloat16 -> widen through its raw bits
            a = (a.view(np.uint16).astype(np.uint32) << 16).view(np.float32)
        tensors[k] = torch.from_numpy(np.ascontiguousarray(a)).to(getattr(torch, dtype))
    if config is not None:
        from lap_amd.config import get_config
        from lap_amd.params import reference_shapes
        from lap_amd.train import validate_loaded_params

        validate_loaded_params(reference_shapes(get_config(config).model), tensors, allow_partial=allow_partial)
    out = pathlib.Path(out_dir)
    out.mkdir(parents=True, exist_ok=True)
    f = out / "params.safetensors"
    save_file({"params/" + k: v.contiguous() for k, v in tensors.items()}, str(f),
              metadata={"format": "lap reference tree, flattened with '/'", "source": "orbax", "dtype": dtype})
    return f


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("orbax_params_dir")
    ap.add_argument("out_dir")
    ap.add_argument("--config", default=None, help="validate against this TrainConfig's model (e.g. lap, lap_libero)")
    ap.add_argument("--dtype", default="float32", choices=["float32", "bfloat16"])
    ap.add_argument("--allow-partial", action="store_true")
    a = ap.parse_args(argv)
    flat = flatten_tree(restore_orbax(a.orbax_params_dir))
    f = write_params(flat, a.out_dir, config=a.config, dtype=a.dtype, allow_partial=a.allow_partial)
    print(f"wrote {f} ({len(flat)} arrays)")


if __name__ == "__main__":
    main()
