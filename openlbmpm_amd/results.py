"""Result files with the reference's HDF5 group / dataset names
(RKD2Q9.py:348-357, :938-957; ShanChenD2Q9.py:503-511, :940-955).

Backend: PyTables or h5py when importable (same file layout as the reference's `tables` calls);
else the HDF5 C library itself through ctypes (openlbmpm_amd/_hdf5.py: still a real .h5 with the
same groups and datasets); only when no HDF5 is present at all a `.npz` whose keys are the HDF5
paths (`/FluidMacro/FluidDensityRin0`, ...), so that post-processing only has to swap the loader.
`LBMPM_RESULT_BACKEND=tables|h5py|libhdf5|npz` forces one.
"""
import os

import numpy as np


def _backend():
    forced = os.environ.get("LBMPM_RESULT_BACKEND")
    if forced:
        if forced not in ("tables", "h5py", "libhdf5", "npz"):
            raise ValueError("LBMPM_RESULT_BACKEND must be tables, h5py, libhdf5 or npz")
        return forced
    try:
        import tables  # noqa: F401
        return "tables"
    except ImportError:
        pass
    try:
        import h5py  # noqa: F401
        return "h5py"
    except ImportError:
        pass
    from . import _hdf5
    return "libhdf5" if _hdf5.available() else "npz"


class ResultFile:
    def __init__(self, directory, name, groups):
        os.makedirs(directory, exist_ok=True)
        self.backend = _backend()
        self.groups = list(groups)
        self.path = os.path.join(directory, name + (".h5" if self.backend != "npz" else ".npz"))
        self._npz = {}
        if self.backend == "npz":
            import warnings
            import zipfile
            warnings.warn("no HDF5 library found (PyTables, h5py, libhdf5): results go to %s" % self.path)
            zipfile.ZipFile(self.path, "w").close()
        if self.backend == "tables":
            import tables as tb
            f = tb.open_file(self.path, "w")
            for g, title in self.groups:
                f.create_group(f.root, g, title)
            f.close()
        elif self.backend == "h5py":
            import h5py
            with h5py.File(self.path, "w") as f:
                for g, _ in self.groups:
                    f.create_group(g)
        elif self.backend == "libhdf5":
            from . import _hdf5
            _hdf5.create(self.path, self.groups)

    def write(self, group, name, array):
        array = np.asarray(array)
        if self.backend == "tables":
            import tables as tb
            f = tb.open_file(self.path, "a")
            f.create_array("/" + group, name, array)
            f.close()
        elif self.backend == "h5py":
            import h5py
            with h5py.File(self.path, "a") as f:
                f["/%s/%s" % (group, name)] = array
        elif self.backend == "libhdf5":
            from . import _hdf5
            _hdf5.write(self.path, "/%s/%s" % (group, name), array)
        else:
            # no HDF5 anywhere: a zip archive of .npy members (what np.load reads as .npz), one member appended
            # per record -- never the whole history rewritten
            import io
            import zipfile
            key = "/%s/%s" % (group, name)
            if key in self._npz:
                raise ValueError("record %s written twice" % key)
            self._npz[key] = True
            buf = io.BytesIO()
            np.save(buf, array)
            with zipfile.ZipFile(self.path, "a", zipfile.ZIP_DEFLATED) as z:
                z.writestr(key + ".npy", buf.getvalue())


def load_results(path):
    """dict path -> array for either backend"""
    if path.endswith(".npz"):
        d = np.load(path)
        return {k: d[k] for k in d.files}
    try:
        import h5py
        out = {}
        with h5py.File(path, "r") as f:
            f.visititems(lambda n, o: out.__setitem__("/" + n, o[()]) if hasattr(o, "shape") else None)
        return out
    except ImportError:
        pass
    try:
        import tables as tb
    except ImportError:
        from . import _hdf5
        return _hdf5.read_all(path)
    else:
        out = {}
        f = tb.open_file(path, "r")
        for node in f.walk_nodes("/", "Array"):
            out[node._v_pathname] = node.read()
        f.close()
        return out


def read_planes(path, key, z0=None, n=None):
    """dataset `key` ("/group/name") of a result file, or its planes [z0, z0 + n) along the first axis -- read as a slab where the
    backend can (a rank of a distributed run takes its own planes of a 512^3 record, not the record); KeyError when absent"""
    sl = slice(None) if z0 is None else slice(int(z0), int(z0) + int(n))
    if path.endswith(".npz"):
        d = np.load(path)
        if key not in d.files:
            raise KeyError(key)
        return d[key][sl]
    try:
        import h5py
        with h5py.File(path, "r") as f:
            return np.asarray(f[key][sl])
    except ImportError:
        pass
    try:
        import tables as tb
    except ImportError:
        from . import _hdf5
        return _hdf5.read_planes(path, key, None if z0 is None else int(z0), None if z0 is None else int(n))
    f = tb.open_file(path, "r")
    try:
        try:
            node = f.get_node(key)
        except tb.NoSuchNodeError:
            raise KeyError(key)
        return np.asarray(node[sl])
    finally:
        f.close()


def find_result_file(directory, stem):
    """<directory>/<stem>.h5 or .npz, else None"""
    for ext in (".h5", ".npz"):
        p = os.path.join(directory, stem + ext)
        if os.path.isfile(p):
            return p
    return None


class SimulationDiverged(FloatingPointError):
    """a recorded field holds NaN or Inf (the reference lets them propagate silently into its result files)"""


class RecordGuard:
    """What every driver does at its output cadence besides writing: a finiteness check of the recorded fields
    (`nan_guard` = 'raise' (default) | 'warn' | 'off'; environment LBMPM_NAN_GUARD overrides) and ONE logging line
    (logger 'openlbmpm_amd', level INFO) with the record index, the step, lattice updates per second since the last
    record and the sums the caller passes (masses, saturation)."""

    def __init__(self, name, fluid_nodes, nan_guard="raise", group=None, collective=False, device=None):
        """collective: the run is distributed (one slab per process) -- the verdict of a record is then agreed on by all ranks
        (one MAX all-reduce of a flag over `group`) before anyone raises: a rank that raised alone would leave its neighbours
        waiting in the next halo exchange until the transport's watchdog fires"""
        import logging
        import time
        self.name, self.fluid_nodes = name, int(fluid_nodes)
        self.mode = os.environ.get("LBMPM_NAN_GUARD", nan_guard)
        if self.mode not in ("raise", "warn", "off"):
            raise ValueError("nan_guard must be 'raise', 'warn' or 'off'")
        self.log = logging.getLogger("openlbmpm_amd")
        self._clock, self._t, self._step = time.perf_counter, time.perf_counter(), 0
        self.group, self.collective = group, bool(collective)
        self.device = device            # the slab's GPU: NCCL reduces on it, whatever the process's current device is

    def _anyone_bad(self, bad):
        """MAX over the ranks of this rank's 'a field is not finite' flag"""
        if not self.collective:
            return bad
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return bad
        dev = "cpu"
        if dist.get_backend(self.group) == "nccl":
            dev = torch.device("cuda", self.device if self.device is not None else torch.cuda.current_device())
        t = torch.tensor([1 if bad else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(int(t.item()))

    def __call__(self, record, step, fields, sums=None):
        if self.mode != "off":
            msg = None
            for key, a in fields.items():
                if not np.isfinite(a).all():
                    bad = int(a.size - np.isfinite(a).sum())
                    msg = "%s: record %d (step %d): %s holds %d non-finite values" % (self.name, record, step, key, bad)
                    break
            if self._anyone_bad(msg is not None):
                if msg is None:
                    msg = "%s: record %d (step %d): another rank's slab holds non-finite values" % (self.name, record, step)
                if self.mode == "raise":
                    raise SimulationDiverged(msg)
                import warnings
                warnings.warn(msg)
        now = self._clock()
        rate = (step - self._step) * self.fluid_nodes / max(now - self._t, 1e-12) / 1e6 if step > self._step else 0.0
        self._t, self._step = now, step
        if self.log.isEnabledFor(20):
            extra = "".join("  %s %.10g" % kv for kv in (sums or {}).items())
            self.log.info("%s record %d step %d  %.0f MLUPS%s", self.name, record, step, rate, extra)
