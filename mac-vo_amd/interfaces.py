"""Mirror of the reference's plugin boundary so the HIP plugins drop into ``Odometry/MACVO.py``.

When the real MAC-VO checkout is importable (its ``Module`` package and all its third-party deps are present) the
plugin classes in :mod:`macvo_amd.plugins` subclass the reference's own ABCs — that is what makes them visible to
``SubclassRegistry.instantiate`` by name (``Utility/Extensions/SubclassRegistry.py:24-48``) with **no change** to
``Odometry/MACVO.py``.  When it is not importable (this repository's tests, the GPU box) the same names are provided by
the light-weight mirrors below, which keep the reference's contract:

* ``Interface.instantiate(type_name, args)`` / ``Interface.get_class`` / auto-registration under ``cls.name()`` in every
  registry-ancestor, duplicate names rejected (``SubclassRegistry.py:8-48``);
* ``Interface.is_valid_config(cfg)`` dispatching on ``cfg.type`` / ``cfg.args`` and ``_enforce_config_spec`` with the
  exact-key-set rule (``Utility/Extensions/Testable.py:10-41``, ``Utility/Extensions/__init__.py:10-16``);
* the typed records ``IStereoDepth.Output`` (``Module/Frontend/StereoDepth.py:33-40``) and ``IMatcher.Output``
  (``Module/Frontend/Matching.py:21-40``);
* the abstract methods of ``IFrontend`` (``Module/Frontend/Frontend.py:38-118``), ``IKeypointSelector``
  (``Module/KeypointSelector.py:17-48``), ``ICovariance2to3`` (``Module/Covariance/Project2to3.py:16-44``) and
  ``IOptimizer`` (``Module/Optimization/Interface.py:40-241``).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Any, Callable, Generic, TypeVar

import torch

USING_REFERENCE = False
try:  # pragma: no cover - only true inside a full MAC-VO environment
    import Module as _RefModule  # type: ignore
    from Module.Optimization.TwoFramePGO.Graphs import GraphInput as _RefGraphInput, GraphOutput as _RefGraphOutput  # type: ignore

    USING_REFERENCE = True
except Exception:  # noqa: BLE001 - any missing dependency means "not inside the reference"
    _RefModule = None


# ----------------------------------------------------------------------------------------------- registry + config
class PluginRegistry:
    """Name -> class lookup shared along the inheritance chain (behaviour of the reference's SubclassRegistry)."""

    _registry: dict[str, type] = {}

    @classmethod
    def name(cls) -> str:
        return cls.__name__

    def __init_subclass__(cls, **kwargs) -> None:
        super().__init_subclass__(**kwargs)
        bases = [b for b in cls.__bases__ if isinstance(b, type) and issubclass(b, PluginRegistry)]
        if len(bases) != 1:
            raise AssertionError("Does not support diamond inheritance in SubclassRegistry")
        cls._registry = {"": cls}  # each class owns the table of ITS descendants
        for parent in cls.__mro__[1:]:
            if not (isinstance(parent, type) and issubclass(parent, PluginRegistry)):
                continue
            table = parent.__dict__.get("_registry")
            if table is None:
                continue
            if cls.name() in table:
                raise NameError(f"more than one descendant of '{parent.__name__}' is named {cls.name()}")
            table[cls.name()] = cls

    @classmethod
    def get_class(cls, type: str):  # noqa: A002 - the reference's parameter name
        table = cls.__dict__.get("_registry", {})
        if type in table:
            return table[type]
        raise KeyError(f"Get '{type}' from class {cls.__name__}, expect to be one of {list(table.keys())}")

    @classmethod
    def instantiate(cls, type: str, *args, **kwargs):  # noqa: A002
        return cls.get_class(type)(*args, **kwargs)


class ConfigCheck:
    @classmethod
    def is_valid_config(cls, config: SimpleNamespace | None) -> None:
        return None

    @staticmethod
    def _enforce_config_spec(config: Any, spec: dict | Callable[[Any], bool], allow_excessive_cfg: bool = False) -> None:
        if not isinstance(spec, dict):
            if not spec(config):
                raise ValueError(f"Config does not match specification! ({config} does not pass test)")
            return
        if not isinstance(config, SimpleNamespace):
            raise AssertionError(f"expected a namespace for spec keys {list(spec)}, got {config!r}")
        have = vars(config)
        for key, rule in spec.items():
            if key not in have:
                raise KeyError(f"Config does not match specification! (expect to have key {key} but did not found)")
            ConfigCheck._enforce_config_spec(have[key], rule)
        if not allow_excessive_cfg and len(have) != len(spec):
            raise KeyError(f"Excessive Keys: {set(have) - set(spec)} from {list(spec)}")


class ConfigurablePlugin(PluginRegistry, ConfigCheck):
    """= the reference's ConfigTestableSubclass: is_valid_config on an interface dispatches on cfg.type."""

    @classmethod
    def is_valid_config(cls, config: SimpleNamespace | None) -> None:
        assert config is not None
        if not hasattr(config, "type") or not hasattr(config, "args"):
            raise ValueError(f"Unable to dynamically delegate a subclass to test provided config. {config=}")
        cls.get_class(config.type).is_valid_config(config.args)


def _is_device(s) -> bool:
    return isinstance(s, str) and (("cuda" in s) or (s == "cpu"))


# ----------------------------------------------------------------------------------------------- typed records
if USING_REFERENCE:  # pragma: no cover
    IStereoDepth = _RefModule.IStereoDepth
    IMatcher = _RefModule.IMatcher
    IFrontend = _RefModule.IFrontend
    IKeypointSelector = _RefModule.IKeypointSelector
    ICovariance2to3 = _RefModule.ICovariance2to3
    IOptimizer = _RefModule.IOptimizer
    GraphInput, GraphOutput = _RefGraphInput, _RefGraphOutput
else:

    class IStereoDepth(ABC, ConfigurablePlugin):
        @dataclass
        class Output:
            depth: torch.Tensor                                  # B x 1 x H x W float32
            disparity: torch.Tensor | None = None
            cov: torch.Tensor | None = None
            mask: torch.Tensor | None = None                     # bool
            disparity_uncertainty: torch.Tensor | None = None

        def __init__(self, config: SimpleNamespace):
            self.config = config

        @property
        @abstractmethod
        def provide_cov(self) -> bool: ...

        @abstractmethod
        def estimate(self, frame) -> "IStereoDepth.Output": ...

    class IMatcher(ABC, ConfigurablePlugin):
        @dataclass
        class Output:
            flow: torch.Tensor                                   # B x 2 x H x W float32
            cov: torch.Tensor | None = None                      # B x 3 x H x W (uu, vv, uv)
            mask: torch.Tensor | None = None

            @classmethod
            def from_partial_cov(cls, flow, cov, mask=None):
                B, C, H, W = cov.shape
                assert C == 2
                return cls(flow=flow, cov=torch.cat([cov, torch.zeros((B, 1, H, W)).to(cov)], dim=1), mask=mask)

        def __init__(self, config: SimpleNamespace):
            self.config = config

        @property
        @abstractmethod
        def provide_cov(self) -> bool: ...

        @abstractmethod
        def forward(self, frame_t1, frame_t2) -> "IMatcher.Output": ...

        def estimate(self, frame_t1, frame_t2) -> "IMatcher.Output":      # Matching.py:68-70
            with torch.no_grad(), torch.inference_mode():
                return self.forward(frame_t1, frame_t2)

    class IFrontend(ABC, ConfigurablePlugin):
        def __init__(self, config: SimpleNamespace):
            self.config = config

        @property
        @abstractmethod
        def provide_cov(self) -> tuple[bool, bool]: ...

        @abstractmethod
        def estimate_pair(self, frame_t1, frame_t2): ...

        @abstractmethod
        def estimate_depth(self, frame): ...

        def estimate_triplet(self, frame_t1, frame_t2):
            depth_t1 = self.estimate_depth(frame_t1)
            depth_t2, match_t12 = self.estimate_pair(frame_t1, frame_t2)
            return depth_t1, depth_t2, match_t12

        @staticmethod
        def retrieve_pixels(pixel_uv, scalar_map, interpolate: bool = False):
            if scalar_map is None:
                return None
            if interpolate:
                raise NotImplementedError("Not implemented yet")
            return scalar_map[0, ..., pixel_uv[..., 1].long(), pixel_uv[..., 0].long()]

    class IKeypointSelector(ABC, ConfigurablePlugin):
        def __init__(self, config: SimpleNamespace):
            self.config = config

        @abstractmethod
        def select_point(self, frame, numPoint: int, depth0_est, depth1_est, match_est) -> torch.Tensor: ...

    class ICovariance2to3(ABC, ConfigurablePlugin):
        def __init__(self, config: SimpleNamespace):
            self.config = config

        @abstractmethod
        def estimate(self, frame, kp, depth_est, depth_cov, flow_cov) -> torch.Tensor: ...

    T_In, T_Ctx, T_Out = TypeVar("T_In"), TypeVar("T_Ctx"), TypeVar("T_Out")

    class IOptimizer(ABC, Generic[T_In, T_Ctx, T_Out], ConfigurablePlugin):
        """Caller protocol of the reference (``Optimization/Interface.py:40-241``): per frame
        ``write_map(map)`` (join the previous job) ... ``start_optimize(get_graph_data(map, idx))``; ``terminate()``.
        The reference's ``parallel: true`` spawns a CPU child process to overlap the solve with the next frame's
        network; a GPU solver overlaps by running on its own HIP stream instead (see HIP_TwoFrame_PGO)."""

        def __init__(self, config: SimpleNamespace) -> None:
            self.config = config
            self.is_parallel_mode = bool(config.parallel)
            self.context = self.init_context(config)
            self.optimize_res = None
            self.has_opt_job = False

        @staticmethod
        @abstractmethod
        def init_context(config): ...

        @staticmethod
        @abstractmethod
        def _optimize(context, graph_data): ...

        def get_graph_data(self, global_map, frame_idx, observations=None, edges=None):
            raise NotImplementedError

        def write_graph_data(self, result, global_map) -> None:
            raise NotImplementedError

        @property
        def is_running(self) -> bool:
            return False

        def start_optimize(self, graph_data) -> None:
            self.has_opt_job = True
            self.context, self.optimize_res = self._optimize(self.context, graph_data)

        def get_result(self):
            return self.optimize_res

        get_optimal = get_result

        def sequential_optimize(self, graph_data):
            return self._optimize(self.context, graph_data)[1]

        def write_map(self, global_map) -> None:
            self.write_graph_data(self.get_result(), global_map)

        def terminate(self) -> None:
            return None

    @dataclass
    class GraphInput:                         # Module/Optimization/TwoFramePGO/Graphs.py:11-21
        frame_idx: torch.Tensor
        from_idx: torch.Tensor
        init_motion: torch.Tensor             # SE3 [1,7]
        baseline: torch.Tensor
        observations: Any                     # MatchObs bundle: .data[...] per Module/Map/Template.py
        points: Any                           # PointNode bundle
        images_intrinsic: torch.Tensor        # [3,3]
        edges_index: torch.Tensor
        device: str

    @dataclass
    class GraphOutput:                        # Graphs.py:24-28
        motion: torch.Tensor
        from_idx: torch.Tensor
        frame_idx: torch.Tensor
