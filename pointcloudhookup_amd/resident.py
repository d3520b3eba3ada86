"""Stage A -> stage B hand-off without re-reading the intermediate LAS file.

In the reference the voxel stage writes ``output/point_2.las`` (ui/import_PC.py:61-65,
pyGUI_towers_test.py:344-368) and tower extraction reads it back (utils/tower_extraction.py:60-62): on a
100 M-point cloud that file is 2.9 GB, and writing + re-reading it is more than half of the wall clock of the
pair.  The file must exist (drop-in parity: other readers use it), but ``extract_towers`` need not read it:
``run_voxel_downsampling`` registers the int32 records it has just laid out - they are still on the device -
under the file's real path; ``extract_towers`` called on that path takes them instead of the file, provided the
file on disk is still the one that was written (size and mtime_ns as recorded when the writer finished).  Any
mismatch - the file was touched, replaced, truncated, removed - and the file is read as before.

The writer may run in the background (``import_PC.ASYNC_WRITE`` / PCH_ASYNC_LAS_WRITE=1; off by default, because
an unchanged caller may open the file with its own reader as soon as ``run_voxel_downsampling`` returns):
``take`` then hands the records out at once and ``settle`` - called by extract_towers before it returns - joins
the writer and checks the file.

One entry (the latest file written); consumed by the first ``take``.  PCH_RESIDENT_HANDOFF=0 switches it off.
"""
from __future__ import annotations

import os
import threading

_LOCK = threading.Lock()
_ENTRY = None


def enabled():
    return os.environ.get("PCH_RESIDENT_HANDOFF", "1") != "0"


class Entry:
    def __init__(self, path, header, records):
        self.path = os.path.realpath(path)
        self.header = header            # las.LasHeader as written (point_format, version, scales, offsets)
        self.records = records          # int32 [n,3] device tensor
        self.writer = None              # background thread, or None when the file was written synchronously
        self.error = None               # exception of the background writer
        self.stamp = None               # (size, mtime_ns) of the finished file

    def finished(self):
        return self.writer is None or not self.writer.is_alive()

    def join(self):
        if self.writer is not None:
            self.writer.join()

    def stamp_now(self):
        st = os.stat(self.path)
        self.stamp = (st.st_size, st.st_mtime_ns)

    def file_matches(self):
        """the file on disk is the one the writer finished"""
        if self.error is not None or self.stamp is None:
            return False
        try:
            st = os.stat(self.path)
        except OSError:
            return False
        return (st.st_size, st.st_mtime_ns) == self.stamp


def register(entry):
    global _ENTRY
    if not enabled():
        return
    with _LOCK:
        old, _ENTRY = _ENTRY, entry
    if old is not None and old is not entry:
        old.join()                      # never leave a writer behind unobserved
        old.records = None


def take(path):
    """The entry registered for ``path`` if its records may stand in for the file: the writer is still running
    (nobody else can have a finished file yet; ``settle`` re-checks) or the file still carries the writer's stamp.
    The entry is consumed either way."""
    global _ENTRY
    if not enabled():
        return None
    real = os.path.realpath(path)
    with _LOCK:
        e = _ENTRY
        if e is None or e.path != real:
            return None
        _ENTRY = None
    if e.finished():
        e.join()
        if not e.file_matches():
            e.records = None
            return None
    return e


def settle(entry):
    """Joins the background writer (if any) and says whether the file on disk is what the records were: False
    means the caller's result must not stand (the file was changed while it was being written, or writing failed)."""
    entry.join()
    entry.records = None
    return entry.file_matches()


def wait_for_writers():
    """Blocks until no LAS writer runs in the background (tests, interpreter exit)."""
    with _LOCK:
        e = _ENTRY
    if e is not None:
        e.join()
