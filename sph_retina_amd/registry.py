"""Registry shim: register into mmdet's registries when mmdet is importable (so `build_iou_calculator` /
`build_loss` find our classes by the reference's type strings), otherwise into a local Registry with the same
`register_module()` / `build()` surface (mmdet/core/bbox/iou_calculators/builder.py:4-9)."""


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._modules[key] = cls
            return cls
        return _reg(module) if module is not None else _reg

    def get(self, key):
        return self._modules.get(key)

    def build(self, cfg, default_args=None):
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        typ = args.pop('type')
        cls = self.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f'{typ} is not in the {self.name} registry')
        return cls(**args)

    def __contains__(self, key):
        return key in self._modules


def _mmdet_registry(path, attr, local_name):
    try:
        mod = __import__(path, fromlist=[attr])
        return getattr(mod, attr), True
    except Exception:  # mmdet / mmcv not installed (e.g. this container): local registry
        return Registry(local_name), False


IOU_CALCULATORS, IOU_CALCULATORS_IS_MMDET = _mmdet_registry('mmdet.core.bbox.iou_calculators.builder',
                                                            'IOU_CALCULATORS', 'iou_calculator')
LOSSES, LOSSES_IS_MMDET = _mmdet_registry('mmdet.models.builder', 'LOSSES', 'loss')
BBOX_ASSIGNERS, BBOX_ASSIGNERS_IS_MMDET = _mmdet_registry('mmdet.core.bbox.builder', 'BBOX_ASSIGNERS', 'bbox_assigner')
BBOX_CODERS, BBOX_CODERS_IS_MMDET = _mmdet_registry('mmdet.core.bbox.builder', 'BBOX_CODERS', 'bbox_coder')


def build_iou_calculator(cfg, default_args=None):
    return IOU_CALCULATORS.build(cfg, default_args) if not IOU_CALCULATORS_IS_MMDET else \
        __import__('mmdet.core.bbox.iou_calculators.builder', fromlist=['x']).build_iou_calculator(cfg, default_args)


def build_loss(cfg):
    return LOSSES.build(cfg) if not LOSSES_IS_MMDET else \
        __import__('mmdet.models.builder', fromlist=['x']).build_loss(cfg)


def build_assigner(cfg):
    return BBOX_ASSIGNERS.build(cfg) if not BBOX_ASSIGNERS_IS_MMDET else \
        __import__('mmdet.core.bbox.builder', fromlist=['x']).build_assigner(cfg)


def build_bbox_coder(cfg):
    return BBOX_CODERS.build(cfg) if not BBOX_CODERS_IS_MMDET else \
        __import__('mmdet.core.bbox.builder', fromlist=['x']).build_bbox_coder(cfg)
