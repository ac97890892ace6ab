"""Container-only loader for the *unmodified* reference Python (TEST INFRASTRUCTURE, never shipped).

Loads the torch-only modules of the reference from /root/reference by file path so that
``oracle/gen_goldens.py`` can emit the committed fixtures under ``tests/golden/`` and so that
``tests/test_oracle_golden.py`` can check the C / numpy restatements against them.  Nothing here runs on the GPU box (the reference does not travel); every entry point
refuses to run when /root/reference is missing.

Recipe: the reference's ``sphdet/iou/sph_iou_api.py:2`` imports the un-vendored
``mmcv.ops`` (mmcv-full 1.6.0).  mmcv is not installable here, so ``mmcv.ops.box_iou_rotated`` and
``mmcv.ops.diff_iou_rotated_2d`` are bound to the reference's OWN vendored pure-torch
``sphdet/iou/diff_iou_rotated.py`` (declared by its header to be the bug-fixed form of that op).
Goldens produced through this path therefore pin: jitter + transform + vendored planar IoU.  The
mmcv C++/CUDA kernel itself stays un-pinned beyond the reference's own ``mean|Δ| < 1e-6`` assertion
(``tests/test_sph_iou_loss.py:34``).
"""
import importlib.util
import os
import sys
import types

REF = '/root/reference'


def available():
    return os.path.isdir(os.path.join(REF, 'sphdet', 'iou'))


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _pkg(name, relpath):
    mod = types.ModuleType(name)
    mod.__path__ = [os.path.join(REF, relpath)]
    sys.modules[name] = mod
    return mod


_CACHE = {}


def load_reference():
    """Returns a namespace with the reference callables for the hot path."""
    if 'ns' in _CACHE:
        return _CACHE['ns']
    if not available():
        raise RuntimeError('/root/reference is not mounted: the reference oracle is container-only')
    sys.dont_write_bytecode = True
    import torch  # noqa: F401

    for name, rel in [('sphdet', 'sphdet'), ('sphdet.iou', 'sphdet/iou'), ('sphdet.bbox', 'sphdet/bbox'),
                      ('sphdet.losses', 'sphdet/losses'), ('sphdet.bbox.nms', 'sphdet/bbox/nms')]:
        _pkg(name, rel)

    diff = _load('sphdet.iou.diff_iou_rotated', 'sphdet/iou/diff_iou_rotated.py')

    def box_iou_rotated(b1, b2, mode='iou', aligned=False, clockwise=True):
        assert aligned
        corners1 = diff.box2corners(b1[None])
        corners2 = diff.box2corners(b2[None])
        inter, _ = diff.oriented_box_intersection_2d(corners1, corners2)
        inter = inter[0]
        a1 = b1[:, 2] * b1[:, 3]
        a2 = b2[:, 2] * b2[:, 3]
        return inter / (a1 + a2 - inter) if mode == 'iou' else inter / a1

    mmcv = types.ModuleType('mmcv')
    mmcv.__version__ = '1.6.0'
    mmcv.jit = lambda **k: (lambda f: f)
    ops = types.ModuleType('mmcv.ops')
    ops.box_iou_rotated = box_iou_rotated
    ops.diff_iou_rotated_2d = diff.diff_iou_rotated_2d
    ops.bbox_overlaps = None
    ops.batched_nms = None
    mmcv.ops = ops
    sys.modules['mmcv'] = mmcv
    sys.modules['mmcv.ops'] = ops

    kf = types.ModuleType('sphdet.bbox.kent_formator')
    kf.deg2kent = None
    sys.modules['sphdet.bbox.kent_formator'] = kf
    kc = types.ModuleType('sphdet.iou.kent_iou_calculator')
    kc.kent_iou_calculator = None
    sys.modules['sphdet.iou.kent_iou_calculator'] = kc

    box_formator = _load('sphdet.bbox.box_formator', 'sphdet/bbox/box_formator.py')
    if not hasattr(box_formator, 'Planar2KentTransform'):
        box_formator.Planar2KentTransform = None
    std = _load('sphdet.iou.sph2pob_standard', 'sphdet/iou/sph2pob_standard.py')
    eff = _load('sphdet.iou.sph2pob_efficient', 'sphdet/iou/sph2pob_efficient.py')
    leg = _load('sphdet.iou.sph2pob_legacy', 'sphdet/iou/sph2pob_legacy.py')
    _load('sphdet.iou.approximate_ious', 'sphdet/iou/approximate_ious.py')
    _load('sphdet.iou.unbiased_iou_bfov', 'sphdet/iou/unbiased_iou_bfov.py')
    _load('sphdet.iou.unbiased_iou_rbfov', 'sphdet/iou/unbiased_iou_rbfov.py')
    api = _load('sphdet.iou.sph_iou_api', 'sphdet/iou/sph_iou_api.py')
    for k in ('unbiased_iou', 'sph2pob_standard_iou', 'sph2pob_legacy_iou', 'sph2pob_efficient_iou',
              'naive_iou', 'fov_iou', 'sph_iou'):
        setattr(sys.modules['sphdet.iou'], k, getattr(api, k))

    # losses: real vendored mmdet/models/losses/utils.py + tiny registries
    class _Reg:
        def register_module(self, *a, **k):
            return lambda c: c

    for name in ['mmdet', 'mmdet.models', 'mmrotate', 'mmrotate.models']:
        sys.modules.setdefault(name, types.ModuleType(name))
    b = types.ModuleType('mmdet.models.builder')
    b.LOSSES = _Reg()
    sys.modules['mmdet.models.builder'] = b
    lutils = _load('mmdet.models.losses.utils', 'mmdet/models/losses/utils.py')
    ml = types.ModuleType('mmdet.models.losses')
    ml.weighted_loss = lutils.weighted_loss
    sys.modules['mmdet.models.losses'] = ml
    ml.__path__ = [os.path.join(REF, 'mmdet', 'models', 'losses')]
    ml.L1Loss = _load('mmdet.models.losses.smooth_l1_loss', 'mmdet/models/losses/smooth_l1_loss.py').L1Loss
    rl = types.ModuleType('mmrotate.models.losses')
    import torch.nn as nn
    rl.RotatedIoULoss = nn.Module
    rl.GDLoss = nn.Module
    rl.KFLoss = nn.Module
    sys.modules['mmrotate.models.losses'] = rl
    tr = _load('sphdet.losses.sph2pob_transform', 'sphdet/losses/sph2pob_transform.py')
    il = _load('sphdet.losses.sph2pob_iou_loss', 'sphdet/losses/sph2pob_iou_loss.py')
    l1 = _load('sphdet.losses.sph2pob_l1_loss', 'sphdet/losses/sph2pob_l1_loss.py')
    nms = _load('sphdet.bbox.nms.sph_nms', 'sphdet/bbox/nms/sph_nms.py')
    gen = _load('ref_tests_generate_data', 'tests/utils/generate_data.py')
    # box coders: the vendored mmdet base class + a no-op registry
    for name in ['mmdet.core', 'mmdet.core.bbox', 'mmdet.core.bbox.coder']:
        sys.modules.setdefault(name, types.ModuleType(name))
    _load('mmdet.core.bbox.coder.base_bbox_coder', 'mmdet/core/bbox/coder/base_bbox_coder.py')
    cb = types.ModuleType('mmdet.core.bbox.builder')
    cb.BBOX_CODERS = _Reg()
    sys.modules['mmdet.core.bbox.builder'] = cb
    _pkg('sphdet.bbox.coder', 'sphdet/bbox/coder')
    coder4 = _load('sphdet.bbox.coder.delta_xywh_sph_bbox_coder', 'sphdet/bbox/coder/delta_xywh_sph_bbox_coder.py')
    coder5 = _load('sphdet.bbox.coder.delta_xywha_rsph_bbox_coder', 'sphdet/bbox/coder/delta_xywha_rsph_bbox_coder.py')

    # MaxIoUAssigner: the vendored mmdet class, unmodified (pure torch); registry + calculator builder stubbed — the
    # calculator handed to the constructor is returned as it is (gen_goldens passes the reference's own IoU function)
    cb.BBOX_ASSIGNERS = _Reg()
    sys.modules.setdefault('mmdet.utils', types.ModuleType('mmdet.utils'))
    sys.modules['mmdet.utils'].util_mixins = _load('mmdet.utils.util_mixins', 'mmdet/utils/util_mixins.py')
    _pkg('mmdet.core.bbox.assigners', 'mmdet/core/bbox/assigners')
    ic = types.ModuleType('mmdet.core.bbox.iou_calculators')
    ic.build_iou_calculator = lambda cfg: cfg
    sys.modules['mmdet.core.bbox.iou_calculators'] = ic
    _load('mmdet.core.bbox.assigners.assign_result', 'mmdet/core/bbox/assigners/assign_result.py')
    _load('mmdet.core.bbox.assigners.base_assigner', 'mmdet/core/bbox/assigners/base_assigner.py')
    assigner = _load('mmdet.core.bbox.assigners.max_iou_assigner', 'mmdet/core/bbox/assigners/max_iou_assigner.py')

    ns = types.SimpleNamespace(
        assigner=assigner, api=api, std=std, eff=eff, leg=leg, diff=diff, box_formator=box_formator, transform=tr, iou_loss=il,
        nms=nms, gen=gen, loss_utils=lutils, coder4=coder4, coder5=coder5, l1_loss=l1)
    _CACHE['ns'] = ns
    return ns
