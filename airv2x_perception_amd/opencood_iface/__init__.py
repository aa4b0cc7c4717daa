"""Host-side mirror of the reference's model interface for the Where2Comm-LiDAR hot path.

Module / class names follow ``opencood.models`` so that the reference's name registry
(tools/train_utils.py:288-325 ``create_model``) resolves them through the one-line binding
shown in INTEGRATION.md.
"""
from .airv2x_where2com import Airv2xWhere2com  # noqa: F401
from .airv2x_cobevt import Airv2xCoBEVT  # noqa: F401,E402
from .airv2x_v2xvit import Airv2xV2XVit  # noqa: F401,E402
from .airv2x_when2com import Airv2xWhen2com  # noqa: F401,E402
from .airv2x_v2vnet import Airv2xV2VNet  # noqa: F401,E402


def create_model(hypes):
    """Harness counterpart of the reference's name registry (tools/train_utils.py:288-325): import the module named by
    ``hypes["model"]["core_method"]`` from THIS package and instantiate the class whose lower-cased name equals
    ``core_method.replace("_", "")`` with ``hypes["model"]["args"]``.  Unknown names raise (the reference prints and exits)."""
    import importlib
    name = hypes["model"]["core_method"]
    try:
        mod = importlib.import_module(f"{__name__}.{name}")
    except ModuleNotFoundError as e:
        raise ValueError(f"core_method {name!r}: no such model in the MI355X build (airv2x_where2com, airv2x_cobevt, airv2x_v2xvit, airv2x_when2com, airv2x_v2vnet)") from e
    target = name.replace("_", "").lower()
    for attr, cls in vars(mod).items():
        if attr.lower() == target and isinstance(cls, type):
            return cls(hypes["model"]["args"])
    raise ValueError(f"module {mod.__name__} has no class named {target} (ignoring case)")


def install_import_shims(force=False):
    """Register the two native modules the reference's callers import before any model runs under the reference's
    own module names, so that its UNMODIFIED ``build_dataset`` / ``build_postprocessor`` import on a ROCm box:
    ``opencood.pcdet_utils.roiaware_pool3d.roiaware_pool3d_cuda`` (intermediate_fusion_dataset.py:24-26 -> roiaware_pool3d_utils.py:5) and
    ``opencood.utils.box_overlaps`` (voxel_postprocessor.py:20).  An already importable module of that name (a CUDA build) is left alone
    unless ``force``.  Returns the names that were registered."""
    import importlib.util
    import sys
    from . import box_overlaps, roiaware_pool3d_cuda
    done = []
    for name, mod in (("opencood.pcdet_utils.roiaware_pool3d.roiaware_pool3d_cuda", roiaware_pool3d_cuda),
                      ("opencood.utils.box_overlaps", box_overlaps)):
        if not force:
            if name in sys.modules:
                continue
            try:
                if importlib.util.find_spec(name) is not None:
                    continue
            except (ImportError, ValueError, AttributeError):
                pass
        sys.modules[name] = mod
        parent, _, leaf = name.rpartition(".")
        if parent in sys.modules:
            setattr(sys.modules[parent], leaf, mod)
        done.append(name)
    return done
