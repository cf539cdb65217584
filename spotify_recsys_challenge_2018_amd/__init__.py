"""MI355X-native DAE scoring / training path."""
import os as _os

# The scoring loop keeps several launches in flight on streams of their own (dae_pipeline_*: lanes + a copy stream; bench.py:
# four contexts per mode).  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that
# share a queue serialise: through DAE.recommend_iter the exact_bf16 loop ran at 3.9 M playlists/s with 16 queues next to
# bench.py's other contexts and at 5.3 M with 32.  Read by the runtime when it initialises (the first HIP call, not the
# import of torch); an exported value wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
