"""Harness that makes the *reference* DiffSound Python importable on a CPU-only box.

Used ONLY by tests/golden/make_golden.py, in the build container where
/root/reference exists.  It never runs on the GPU box and is never imported by
the product package.  It installs three shims and, for the
spectral loss only (``install_spectral``, round 5), three more:

  1. a stand-in ``torch_scatter`` module exposing ``scatter(src, index, dim,
     dim_size, reduce)`` (reference call sites: src/diffelastic/deform.py:165,
     src/diffelastic/mesh.py:176),
  2. a stand-in ``meshio`` module whose ``read()`` parses Gmsh 2.2 binary
     ``.msh`` files (reference call sites: src/diffelastic/mesh.py:38,50,187),
  3. ``Tensor.cuda`` / ``Module.cuda`` as no-ops (the reference hard-codes
     ``.cuda()`` everywhere).
  4. a stand-in ``torchaudio`` whose ``transforms.Spectrogram`` restates the
     published algorithm of torchaudio 2.0.2 (requirements.txt:172;
     ``torchaudio.functional.spectrogram``: torch.stft with a periodic Hann
     window of n_fft samples, centred frames, reflect padding, one-sided,
     not normalised, ``abs().pow(power)`` with power 2) - the reference's call
     site is src/ddsp/mss_loss.py:79-80.  What G9 pins is therefore the
     reference's OWN lines (weights, log2, eps, alpha, hop, the sum over the
     scales) run for real; the Spectrogram underneath is a restatement,
  5. ``torchvision.transforms.functional.gaussian_blur`` and
  6. ``geomloss.SamplesLoss``: imported by the module, never called on the
     'l1_loss' / 'rmse_loss' paths - stand-ins that raise when called.
"""
import struct
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"


def _scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() != src.dim():
        shape = [1] * src.dim()
        shape[dim] = -1
        index = index.reshape(shape).expand_as(src)
    if dim_size is None:
        dim_size = int(index.max()) + 1
    size = list(src.shape)
    size[dim] = dim_size
    if reduce in ("sum", "add"):
        res = torch.zeros(size, dtype=src.dtype, device=src.device)
        return res.scatter_add_(dim, index, src)
    if reduce == "min":
        res = torch.full(size, torch.iinfo(src.dtype).max if not src.dtype.is_floating_point else float("inf"),
                         dtype=src.dtype, device=src.device)
        return res.scatter_reduce_(dim, index, src, reduce="amin", include_self=True)
    raise NotImplementedError(reduce)


def read_gmsh22_binary(path):
    """Parse a Gmsh 2.2 binary file (8-byte reals) -> (points f64 (nv,3), tets i64 (T,4))."""
    data = open(path, "rb").read()
    pos = data.index(b"$Nodes\n") + len(b"$Nodes\n")
    end = data.index(b"\n", pos)
    nv = int(data[pos:end])
    pos = end + 1
    rec = np.dtype([("id", "<i4"), ("xyz", "<f8", 3)])
    nodes = np.frombuffer(data, dtype=rec, count=nv, offset=pos)
    ids = nodes["id"].astype(np.int64)
    pts = np.array(nodes["xyz"], dtype=np.float64)
    pos = data.index(b"$Elements\n") + len(b"$Elements\n")
    end = data.index(b"\n", pos)
    ne = int(data[pos:end])
    pos = end + 1
    tets = []
    done = 0
    while done < ne:
        etype, cnt, ntags = struct.unpack_from("<iii", data, pos)
        pos += 12
        nn = {4: 4, 2: 3, 1: 2, 15: 1, 11: 10}[etype]
        width = 1 + ntags + nn
        block = np.frombuffer(data, dtype="<i4", count=cnt * width, offset=pos).reshape(cnt, width)
        pos += 4 * cnt * width
        if etype == 4:
            tets.append(block[:, 1 + ntags:].astype(np.int64))
        done += cnt
    tets = np.concatenate(tets, axis=0)
    # node ids are 1-based and dense in the fixture files
    lut = np.full(ids.max() + 1, -1, dtype=np.int64)
    lut[ids] = np.arange(nv)
    return pts, lut[tets]


class _Cell:
    def __init__(self, type_, data):
        self.type = type_
        self.data = data


class _Mesh:
    def __init__(self, points, cells):
        self.points = points
        self.cells = [_Cell(t, d) for (t, d) in cells]
        self.cells_dict = {t: d for (t, d) in cells}


def _meshio_read(path):
    pts, tets = read_gmsh22_binary(path)
    return _Mesh(pts, [("tetra", tets)])


def install():
    ts = types.ModuleType("torch_scatter")
    ts.scatter = _scatter
    sys.modules["torch_scatter"] = ts
    mio = types.ModuleType("meshio")
    mio.read = _meshio_read
    mio.Mesh = _Mesh
    mio.write = lambda *a, **k: None
    sys.modules["meshio"] = mio
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


class _Spectrogram(torch.nn.Module):
    """torchaudio.transforms.Spectrogram of torchaudio 2.0.2, restated (constructor defaults and forward of
    torchaudio/transforms/_transforms.py + functional.spectrogram; normalized=False, pad=0 as the reference uses it)."""

    def __init__(self, n_fft=400, win_length=None, hop_length=None, pad=0, window_fn=torch.hann_window, power=2.0,
                 normalized=False, wkwargs=None, center=True, pad_mode="reflect", onesided=True):
        super().__init__()
        assert pad == 0 and not normalized and power is not None
        self.n_fft = n_fft
        self.win_length = win_length if win_length is not None else n_fft
        self.hop_length = hop_length if hop_length is not None else self.win_length // 2
        self.register_buffer("window", window_fn(self.win_length) if wkwargs is None else window_fn(self.win_length, **wkwargs),
                             persistent=False)
        self.power, self.center, self.pad_mode, self.onesided = power, center, pad_mode, onesided

    def forward(self, waveform):
        shape = waveform.size()
        x = waveform.reshape(-1, shape[-1])
        spec = torch.stft(x, self.n_fft, self.hop_length, self.win_length, self.window, self.center, self.pad_mode, False,
                          self.onesided, return_complex=True)
        spec = spec.reshape(shape[:-1] + spec.shape[-2:])
        return spec.abs() if self.power == 1.0 else spec.abs().pow(self.power)


def _never(name):
    def f(*a, **k):
        raise RuntimeError(f"{name}: stand-in of the golden harness - not part of the 'l1_loss' / 'rmse_loss' paths")
    return f


def install_spectral():
    """Shims 4-6: what src/ddsp/mss_loss.py imports at module level."""
    ta, tt = types.ModuleType("torchaudio"), types.ModuleType("torchaudio.transforms")
    tt.Spectrogram = _Spectrogram
    ta.transforms = tt
    sys.modules["torchaudio"], sys.modules["torchaudio.transforms"] = ta, tt
    tv, tvt, tvf = (types.ModuleType(n) for n in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"))
    tvf.gaussian_blur = _never("torchvision.transforms.functional.gaussian_blur")
    tv.transforms, tvt.functional = tvt, tvf
    for m in (tv, tvt, tvf):
        sys.modules[m.__name__] = m
    gl = types.ModuleType("geomloss")

    class SamplesLoss:  # (constructed by SSSLoss.__init__, called on the 'geomloss' path only)
        def __init__(self, *a, **k):
            pass

        __call__ = _never("geomloss.SamplesLoss")

    gl.SamplesLoss = SamplesLoss
    sys.modules["geomloss"] = gl
