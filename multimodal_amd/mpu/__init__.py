"""Model-parallel utilities: same export list as megatron/mpu/__init__.py:17-53 for the part on the hot path."""
from . import mappings  # noqa: F401
from .initialize import (destroy_model_parallel, get_data_parallel_group, get_data_parallel_rank,  # noqa: F401
                         get_data_parallel_src_rank, get_data_parallel_world_size, get_fp32_allreduce,
                         get_model_parallel_group, get_model_parallel_rank, get_model_parallel_src_rank,
                         get_model_parallel_world_size, get_topology, group_layout, initialize_model_parallel,
                         is_unitialized, model_parallel_is_initialized, set_model_parallel_rank,
                         set_model_parallel_world_size)
from .layers import ColumnParallelLinear, RowParallelLinear  # noqa: F401
from .mappings import (copy_to_model_parallel_region, gather_from_model_parallel_region,  # noqa: F401
                       reduce_from_model_parallel_region, scatter_to_model_parallel_region)
from .random import checkpoint, get_cuda_rng_tracker, model_parallel_cuda_manual_seed  # noqa: F401
from .utils import VocabUtility, divide, ensure_divisibility, split_tensor_along_last_dim  # noqa: F401
from .data_parallel import OverlappedGradReducer, allreduce_gradients, plan_buckets  # noqa: F401
from .cross_entropy import vocab_parallel_cross_entropy  # noqa: F401
from .overlap import run_interleaved, split_batch  # noqa: F401
