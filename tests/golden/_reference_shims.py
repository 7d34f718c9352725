"""Stand-in modules that let the reference's hot-path files be imported in the BUILD CONTAINER.

Used only by the golden-vector generators in this directory (never by tests at run time, never
on the GPU box -- /root/reference does not exist there).  The reference's TripoSR package needs
three modules that are not installed here:

  omegaconf  -- only OmegaConf.load / resolve / structured / merge are touched
                (tsr/utils.py:12,16-18 ; tsr/system.py:9,61-62)
  bpy        -- only touched inside TSR.import_obj_blender (system.py:127-168)
  skimage    -- `from skimage import measure` at isosurface.py:7; the goldens for marching
                cubes come from the real scikit-image under /opt/conda (make_mc_goldens.py)

Nothing here restates reference arithmetic.
"""
import dataclasses
import sys
import types

import yaml


class _Cfg(dict):
    """dict with attribute access (enough of DictConfig for the reference)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return _Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return [_wrap(v) for v in x]
    return x


class OmegaConf:
    @staticmethod
    def load(path):
        with open(path) as f:
            return _wrap(yaml.safe_load(f))

    @staticmethod
    def resolve(cfg):
        root = cfg

        def look(path):
            cur = root
            for p in path.split("."):
                cur = cur[p]
            return cur

        def rec(d):
            for k, v in list(d.items()):
                if isinstance(v, dict):
                    rec(v)
                elif isinstance(v, str) and v.startswith("${") and v.endswith("}"):
                    d[k] = look(v[2:-1])

        rec(cfg)

    @staticmethod
    def structured(cls):
        out = _Cfg()
        for f in dataclasses.fields(cls):
            if f.default is not dataclasses.MISSING:
                out[f.name] = f.default
            elif f.default_factory is not dataclasses.MISSING:  # type: ignore
                out[f.name] = f.default_factory()  # type: ignore
        return out

    @staticmethod
    def merge(a, b):
        out = _Cfg(a)
        if b is not None:
            for k, v in dict(b).items():
                out[k] = _wrap(v)
        return out


def install(reference_root="/root/reference"):
    om = types.ModuleType("omegaconf")
    om.OmegaConf = OmegaConf
    om.DictConfig = _Cfg
    sys.modules["omegaconf"] = om
    sys.modules["bpy"] = types.ModuleType("bpy")
    sk = types.ModuleType("skimage")
    sk.measure = types.ModuleType("skimage.measure")
    sys.modules["skimage"] = sk
    sys.modules["skimage.measure"] = sk.measure
    pkg_root = reference_root + "/TripoSR"
    if pkg_root not in sys.path:
        sys.path.insert(0, pkg_root)
