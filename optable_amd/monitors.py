"""Rectangular recorder plane (reference: optable/monitor.py).

`record` is the step right after the trace inside `_single_ray_tracing`
(optical_table.py:145-146): every finished segment is intersected with the monitor's
rectangle, honouring the segment's length (monitor.py:183-193).  On the MI355X path the
intersection pass runs on the device over the trace's segment stream
(`ot_monitor_record_f64`); this class keeps the reference's accessors on top of the hits.
Plots are out of scope.
"""
import numpy as np

from .components import OpticalComponent
from .shapes import Rectangle


class Monitor(OpticalComponent):
    def __init__(self, origin, width, height, **kwargs):
        super().__init__(origin, **kwargs)
        self.width, self.height = width, height
        self.surface = Rectangle(width, height)
        self._edge_color = "orange"
        self._initialize()

    def _initialize(self):
        self._data_raw = []  # (P_local, intensity, t, ray)
        self._sorted_data = []
        self._updated = False
        self._sort_method = None

    def clear(self):
        self._initialize()

    def get_bbox(self):
        return OpticalComponent.get_bbox(self)

    # -- recording -------------------------------------------------------------------------
    def record(self, rays):
        """Record a list of finished `Ray` segments (device pass via the engine)."""
        from .table import record_monitor_hits

        record_monitor_hits(self, rays)

    def _extend(self, hits):
        self._data_raw.extend(hits)
        self._updated = True

    # -- accessors (monitor.py:24-156) -----------------------------------------------------
    @property
    def ndata(self):
        return len(self._data_raw)

    @property
    def raw_yList(self):
        return np.array([d[0][1] for d in self._data_raw])

    @property
    def raw_zList(self):
        return np.array([d[0][2] for d in self._data_raw])

    @property
    def sortYZIndex(self):
        return np.lexsort((self.raw_zList, self.raw_yList))

    @property
    def sortIDindex(self):
        return np.argsort([d[3]._id for d in self._data_raw])

    def get_data(self, sort="YZ"):
        if (not self._sorted_data) or self._updated or self._sort_method != sort:
            order = {"YZ": lambda: self.sortYZIndex, "ID": lambda: self.sortIDindex}.get(sort)
            if order is not None:
                self._sort_method = sort
                self._sorted_data = [self._data_raw[i] for i in order()]
            self._updated = False
        return self._sorted_data

    data = property(lambda self: self.get_data())

    def _column(self, pick, sort):
        if self.ndata == 0:
            return np.array([])
        return np.array([pick(d) for d in self.get_data(sort=sort)])

    def get_rays(self, sort="YZ"):
        return [] if self.ndata == 0 else [d[3] for d in self.get_data(sort=sort)]

    def get_PList(self, sort="YZ"):
        return self._column(lambda d: d[0], sort)

    def get_yList(self, sort="YZ"):
        axis = self.tangent_Y
        return self._column(lambda d: np.dot(d[0], axis), sort)

    def get_zList(self, sort="YZ"):
        axis = self.tangent_Z
        return self._column(lambda d: np.dot(d[0], axis), sort)

    def get_IList(self, sort="YZ"):
        return self._column(lambda d: d[1], sort)

    def get_tList(self, sort="YZ"):
        return self._column(lambda d: d[2], sort)

    def get_directionList(self, sort="YZ"):
        return self._column(lambda d: d[3].direction, sort)

    def get_tYList(self, sort="YZ"):
        return np.dot(self.get_directionList(sort=sort), self.tangent_Y)

    def get_tZList(self, sort="YZ"):
        return np.dot(self.get_directionList(sort=sort), self.tangent_Z)

    rays = property(lambda self: self.get_rays())
    PList = property(lambda self: self.get_PList())
    yList = property(lambda self: self.get_yList())
    zList = property(lambda self: self.get_zList())
    IList = property(lambda self: self.get_IList())
    tList = property(lambda self: self.get_tList())
    directionList = property(lambda self: self.get_directionList())
    tYList = property(lambda self: self.get_tYList())
    tZList = property(lambda self: self.get_tZList())

    def get_ray_i(self, idx):
        return self.rays[idx], [self.yList[idx], self.zList[idx], self.tYList[idx], self.tZList[idx], self.IList[idx]]

    def get_ray_id(self, ray_id):
        for idx, r in enumerate(self.rays):
            if r._id == ray_id:
                return self.get_ray_i(idx)
        return None, None

    def get_waist_distance(self):
        out = []
        for r, t in zip(self.rays, self.tList):
            z = r.distance_to_waist(r.q_at_z(t))
            out.append(-z if np.dot(r.direction, self.normal) > 0 else z)
        return np.array(out)

    def get_beam_waist(self):
        """Gaussian waist of every recorded ray at the monitor (monitor.py:218-225; upstream reads an undefined
        `self.rList` there — `self.rays` is what the loop needs)."""
        return np.array([r.waist(r.q_at_z(t)) for r, t in zip(self.rays, self.tList)])

    def get_delta_pos(self):
        """Spacings of the hits sorted by y (monitor.py:227-238)."""
        y, z = self.yList, self.zList
        if len(y) == 0 or len(z) == 0:
            return np.array([0.0]), np.array([0.0])
        order = np.argsort(y)
        return np.diff(y[order]), np.diff(z[order])

    def _get_hist_y(self):
        return np.histogram(self.yList, bins=30, range=(-self.width / 2, self.width / 2))

    @property
    def std_histy(self):
        """Standard deviation of the 30-bin y histogram (monitor.py:248-253)."""
        counts, bins = self._get_hist_y()
        mean = np.sum(counts * bins[:-1]) / np.sum(counts)
        return np.sqrt(np.sum(counts * bins[:-1] ** 2) / np.sum(counts) - mean**2)

    @property
    def sum_intensity(self):
        return np.sum([d[1] for d in self.get_data()])

    @property
    def avg_intensity(self):
        return np.mean([d[1] for d in self.get_data()])

    def export_rays_npz(self, filename: str):
        print(f"Exporting {self.ndata} rays to {filename} ...")
        np.savez(filename, xList=self.yList, yList=self.zList, tXList=self.tYList, tYList=self.tZList, IList=self.IList)


class MonitorHits:
    """`Monitor.record` results for a SegmentBatch, kept on the device (no Python objects).

    Mirrors the accessors of `Monitor` (monitor.py:33-156): `yList/zList` are the local hit point
    projected on the monitor's LAB tangents (as upstream does), `tYList/tZList` the same projections
    of the segment directions, `IList` the intensities, `tList` the distances.  `sort="YZ"` orders by
    local y then z (np.lexsort((z, y))), `sort="ID"` by input-ray index, `sort=None` keeps
    reference order (input-ray-major, then segment order)."""

    def __init__(self, monitor, segs, slot, P, t):
        import torch

        self.monitor, self.segs = monitor, segs
        ray = segs.ray[slot].long()
        if segs.layout == "slots":  # [k][ray] slots: reference order is ray-major, then k
            order = torch.argsort(ray * (segs.capacity // max(segs.n_rays, 1)) + slot // max(segs.n_rays, 1))
        else:
            order = torch.argsort(ray, stable=True)
        self.slot, self.P, self.t, self.ray = slot[order], P[order], t[order], ray[order]

    def __len__(self):
        return int(self.slot.numel())

    def _order(self, sort):
        import torch

        if sort is None:
            return torch.arange(len(self), device=self.slot.device)
        if sort == "ID":
            return torch.argsort(self.ray, stable=True)
        if sort == "YZ":
            by_z = torch.argsort(self.P[:, 2], stable=True)
            return by_z[torch.argsort(self.P[by_z, 1], stable=True)]
        raise ValueError(f"unknown sort {sort!r}")

    def _axis(self, which):
        import torch

        vec = self.monitor.tangent_Y if which == "Y" else self.monitor.tangent_Z
        return torch.as_tensor(np.asarray(vec, dtype=float), device=self.slot.device)

    def PList(self, sort="YZ"):
        return self.P[self._order(sort)]

    def yList(self, sort="YZ"):
        return self.P[self._order(sort)] @ self._axis("Y")

    def zList(self, sort="YZ"):
        return self.P[self._order(sort)] @ self._axis("Z")

    def tList(self, sort="YZ"):
        return self.t[self._order(sort)]

    def IList(self, sort="YZ"):
        return self.segs.intensity[self.slot[self._order(sort)]]

    def ray_index(self, sort="YZ"):
        return self.ray[self._order(sort)]

    def directionList(self, sort="YZ"):
        import torch

        s = self.slot[self._order(sort)]
        return torch.stack([self.segs.dx[s], self.segs.dy[s], self.segs.dz[s]], dim=1)

    def tYList(self, sort="YZ"):
        return self.directionList(sort) @ self._axis("Y")

    def tZList(self, sort="YZ"):
        return self.directionList(sort) @ self._axis("Z")

    def export_rays_npz(self, filename: str):
        """`Monitor.export_rays_npz` (monitor.py:255-269) from the device tensors: the same five arrays in the
        monitor's default "YZ" order (xList / yList are the hit point along the monitor's Y / Z tangents,
        tXList / tYList the direction along them)."""
        print(f"Exporting {len(self)} rays to {filename} ...")
        arrays = {"xList": self.yList(), "yList": self.zList(), "tXList": self.tYList(), "tYList": self.tZList(), "IList": self.IList()}
        np.savez(filename, **{k: v.double().cpu().numpy() for k, v in arrays.items()})
