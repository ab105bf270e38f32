"""kvmatch -- MI355X-native phase-2 candidate verification for KV-match (host-side binding).

Import name ``kvmatch``; lives in the repo directory ``kv-match_amd/`` (add that directory to sys.path)."""
from ._lib import (KvMatchDeviceError, KvMatchIllegalArgument, KvMatchIndexOutOfBounds, KvMatchIOError, LIB_PATH, OPT_EARLY_ABANDON,  # noqa
                   OPT_STRICT_BOUNDS, OPT_UCR_LOOPS, OPT_IN_FLIGHT, OPT_ED_FORCE, OPT_SERIALIZE_ENQUEUE, OPT_ED_LAST_SHAPE, ed_force, ed_shape,
                   OPT_DP_KERNEL, OPT_ED_KERNEL, OPT_LB_CASCADE, OPT_SORT_VIOLATION, OPT_TIMING, STAT_KERNEL_NS, STAT_PASS_FILTER, STAT_PASS_LB_KEOGH,
                   STAT_DTW_SEQUENTIAL, STAT_PASS_LB_KEOGH2, STAT_PASS_LB_KIM, STAT_RECHECKED, STAT_WINDOWS)
from .engine import phase2_time_estimate_ms, phase2_time_estimate_dp_ms, phase1_should_stop  # noqa
from .engine import DeviceGroup, split_intervals  # noqa
from ._lib import KvMatchShardFailed  # noqa
from .engine import (Answers, IndexBuilder, Job, NormQueryEngine, NormQueryEngineDtw, Phase2Context, QueryEngine, QueryEngineDtw,  # noqa
                     QueryStream, StatisticInfo, cnsm_dtw_order, resolve_rho, sort_and_merge_intervals)
from .ucr import UcrDtwQueryExecutor, UcrEdQueryExecutor  # noqa
