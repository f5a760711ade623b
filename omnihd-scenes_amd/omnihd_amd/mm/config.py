"""Load an mmcv-style python config file (the reference's NewScenes configs are standalone: no
``_base_``) into a plain dict, without mmcv."""
import os


def load_config(path):
    ns = {"__file__": os.path.abspath(path)}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), ns)
    return {k: v for k, v in ns.items() if not k.startswith("__") and not callable(v) and not isinstance(v, type(os))}


def build_detector(model_cfg):
    """model=dict(type='BEVFUSION_depth', ...) -> nn.Module, through the DETECTORS registry."""
    import projects.mmdet3d_plugin  # noqa: F401  (registers the plugin's type names)
    from .registry import DETECTORS
    return DETECTORS.build(model_cfg)
