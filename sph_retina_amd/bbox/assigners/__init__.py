from .max_iou_assigner import AssignResult, SphMaxIoUAssigner, assign_wrt_overlaps, fused_assign  # noqa: F401

__all__ = ['SphMaxIoUAssigner', 'AssignResult', 'assign_wrt_overlaps', 'fused_assign']
