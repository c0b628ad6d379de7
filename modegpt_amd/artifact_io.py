"""save_layer's torch.save off the critical path (reference: src/adapters/model_adapter.py:184-191 writes
layer_<i>_<suffix> synchronously after every layer and stage: 3.3 s of the 7.2 s the three stages take on Llama-3-8B).

`ArtifactWriter.submit` enqueues the device-to-host copies of a layer's tensors into pinned buffers on a copy stream ordered
after the caller's stream and returns; a worker thread waits for the copy event, writes the file under a temporary name and
renames it (a reader never sees a half-written file).  `flush()` returns when every submitted file is in place and re-raises the
first error the worker met.  Everything that READS the artefact directory flushes first: convert_model,
sharding.gather_layer_artifacts, the end of run_modegpt.  Same file names, same {name: bf16 tensor} payload (saved from host
copies: the reference reads them back with map_location, model_adapter.py:213)."""
from __future__ import annotations

import atexit
import os
import queue
import threading
from typing import Dict, Optional

import torch


class ArtifactWriter:
    def __init__(self, max_pending: int = 8):
        self._q: "queue.Queue" = queue.Queue(maxsize=max_pending)      # bounds the pinned host memory held (a few hundred MB each)
        self._thread: Optional[threading.Thread] = None
        self._error: Optional[BaseException] = None
        self._copy_streams: Dict[int, "torch.cuda.Stream"] = {}
        self._lock = threading.Lock()
        atexit.register(self._at_exit)

    def submit(self, path: str, weights: dict) -> None:
        self._raise_pending()
        host, event = {}, None
        devs = {t.device for t in weights.values() if torch.is_tensor(t) and t.is_cuda}
        if devs:
            if len(devs) > 1:
                raise ValueError(f"one artefact's tensors on several devices: {sorted(map(str, devs))}")
            dev = next(iter(devs))
            cs = self._copy_streams.get(dev.index)
            if cs is None:
                cs = self._copy_streams[dev.index] = torch.cuda.Stream(device=dev)
            cs.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(cs):
                for k, t in weights.items():
                    if torch.is_tensor(t) and t.is_cuda:
                        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                        h.copy_(t, non_blocking=True)
                        t.record_stream(cs)             # the caller may drop the tensor as soon as this returns
                        host[k] = h
                    else:
                        host[k] = t
                event = torch.cuda.Event()
                event.record(cs)
        else:
            host = dict(weights)
        self._start()
        self._q.put((path, host, event))

    def flush(self) -> None:
        if self._thread is not None:
            self._q.join()
        self._raise_pending()

    def pending(self) -> int:
        return self._q.unfinished_tasks

    def _raise_pending(self) -> None:
        with self._lock:
            err, self._error = self._error, None
        if err is not None:
            raise RuntimeError(f"writing a layer artefact failed: {err!r}") from err

    def _start(self) -> None:
        if self._thread is None or not self._thread.is_alive():
            self._thread = threading.Thread(target=self._run, name="modegpt-artifact-writer", daemon=True)
            self._thread.start()

    def _run(self) -> None:
        while True:
            path, host, event = self._q.get()
            try:
                if event is not None:
                    event.synchronize()
                tmp = f"{path}.tmp{os.getpid()}"
                torch.save(host, tmp)
                os.replace(tmp, path)
            except BaseException as e:      # noqa: BLE001  (kept for flush / the next submit to raise)
                with self._lock:
                    self._error = self._error or e
            finally:
                del host
                self._q.task_done()

    def _at_exit(self) -> None:
        try:
            self.flush()
        except Exception:                   # noqa: BLE001  (interpreter shutdown: nothing left to tell)
            pass
