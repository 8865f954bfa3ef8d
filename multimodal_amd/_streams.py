"""Cross-stream ordering for lazily built device caches (the transposed-weight cache, the rotary tables, the rng tracker's
device state).  The layer runs on more than one HIP stream (the GPT-J branch streams of model/transformer.py, the sub-batch
streams of mpu/overlap.py), and a cache entry is filled by kernels enqueued on whichever stream touches it first; a consumer
on ANOTHER stream must not read it before those kernels have run.  The creator stamps the entry with an event recorded
behind the filling work; every later user on a different stream waits for that event (a no-op once it has completed)."""
import torch


def stamp(device=None):
    """Record an event on the current stream (behind the work that fills a cache entry).  None without a GPU."""
    if not torch.cuda.is_available():
        return None
    st = torch.cuda.current_stream(device)
    ev = torch.cuda.Event()
    ev.record(st)
    return ev, st.cuda_stream, torch.cuda.is_current_stream_capturing()


def wait(stamp_, device=None):
    """Make the current stream wait for the stamped work if it was enqueued on a different stream."""
    if stamp_ is None:
        return
    ev, sid, captured = stamp_
    cur = torch.cuda.current_stream(device)
    if cur.cuda_stream == sid:
        return
    # an event recorded outside a capture must not be waited for inside one (and the other way round): work enqueued
    # before the capture began has been synchronised by the capture's own entry fence
    if captured == torch.cuda.is_current_stream_capturing():
        cur.wait_event(ev)
