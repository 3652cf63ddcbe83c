"""hyperreel_amd -- MI355X-native forward renderer for HyperReel scenes.

Python host code keeps the reference's render_fn / registry / model-YAML surface
(`render.render_fn_dict`, `models.model_dict`, `config`), all arithmetic runs in
libhyperreel_hip.so (hand-written HIP for gfx950) through a C ABI.
"""
from . import config, plan, scenes  # noqa: F401

__all__ = ['config', 'plan', 'scenes']
