"""Python host over the C ABI, mirroring the reference's effects_chain life-cycle
(effects_chain.h:40-53): build -> run block by block -> drain -> destroy.

* ``EffectsChain``: one stream, numpy buffers on the host (PCIe each call) -- the
  compatibility path, same semantics as ``run_effects_chain`` / ``drain_effects_chain``.
* ``BatchChain``: S independent streams, torch CUDA tensors resident in HBM, launches
  on the current torch stream -- the throughput path.
"""
import numpy as np

from .lib import load_library, last_error


class EffectsChain:
    def __init__(self, chain, fs, channels, directory=None):
        import ctypes as C
        self.L = load_library()
        ofs, och = C.c_int(), C.c_int()
        d = directory.encode() if directory else None
        self.h = self.L.dspamd_chain_build(chain.encode(), fs, channels, d, C.byref(ofs), C.byref(och))
        if not self.h:
            raise ValueError(f"dsp_amd: cannot build chain {chain!r}: {last_error()}")
        self.fs, self.channels, self.ofs, self.ochannels = fs, channels, ofs.value, och.value

    def close(self):
        if getattr(self, "h", None):
            self.L.dspamd_chain_destroy(self.h)
            self.h = None

    __del__ = close

    def effect_names(self):
        return [self.L.dspamd_chain_effect_name(self.h, i).decode() for i in range(self.L.dspamd_chain_n_effects(self.h))]

    def drain_frames(self):
        return self.L.dspamd_chain_drain_frames(self.h)

    def reset(self):
        self.L.dspamd_chain_reset(self.h)

    def run(self, x):
        """One run_effects_chain() call: x [frames, channels] float64 -> [oframes, ochannels]."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        frames = x.shape[0]
        if frames == 0:
            return np.zeros((0, self.ochannels))
        cap = max(self.L.dspamd_chain_max_out_frames(self.h, frames), 1)
        out = np.empty((cap, self.ochannels))
        f = self.L.dspamd_chain_run(self.h, x.ctypes.data, frames, out.ctypes.data, cap)
        if f < 0:
            raise RuntimeError(f"dsp_amd: chain_run failed: {last_error()}")
        return out[:f].copy()

    def drain(self, block=2048):
        """One drain_effects_chain() call; returns None when dry."""
        cap = max(self.L.dspamd_chain_max_out_frames(self.h, block), 1)
        out = np.empty((cap, self.ochannels))
        f = self.L.dspamd_chain_drain(self.h, block, out.ctypes.data, cap)
        if f == -1:
            return None
        if f < 0:
            raise RuntimeError(f"dsp_amd: chain_drain failed: {last_error()}")
        return out[:f].copy()

    def process(self, x, block=2048):
        """Whole stream incl. drain, like the CLI loop (dsp.c:1295-1454)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        outs = [self.run(x[p:p + block]) for p in range(0, x.shape[0], block)]
        while True:
            o = self.drain(block)
            if o is None:
                break
            outs.append(o)
        outs = [o for o in outs if o.shape[0]]
        return np.concatenate(outs) if outs else np.zeros((0, self.ochannels))


class BatchChain:
    """S streams x C channels in HBM; tensors are [S, frames, C] float64 CUDA."""

    def __init__(self, chain, fs, channels, n_streams, max_frames, directory=None, device=None):
        import torch
        self.torch = torch
        self.L = load_library()
        if device is not None:
            torch.cuda.set_device(device)
            self.L.dspamd_set_device(torch.cuda.current_device())
        d = directory.encode() if directory else None
        self.h = self.L.dspamd_batch_create(chain.encode(), fs, channels, n_streams, max_frames, d)
        if not self.h:
            raise ValueError(f"dsp_amd: cannot build batch chain {chain!r}: {last_error()}")
        self.S, self.fs, self.channels, self.max_frames = n_streams, fs, channels, max_frames
        self.ofs = self.L.dspamd_batch_out_fs(self.h)
        self.ochannels = self.L.dspamd_batch_out_channels(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.dspamd_batch_destroy(self.h)
            self.h = None

    __del__ = close

    def plan(self):
        return self.L.dspamd_batch_plan(self.h).decode()

    def drain_frames(self):
        return self.L.dspamd_batch_drain_frames(self.h)

    def max_out_frames(self, frames):
        return self.L.dspamd_batch_max_out_frames(self.h, frames)

    def _stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    def run(self, x, out=None):
        """x: [S, frames, C] cuda float64 -- contiguous, or a view x = buf[:, :frames, :] of a contiguous [S, stride, C] buffer
        (padded slabs: see dspamd_batch_run_strided).  Returns a view [S, oframes, C_out] of `out` (allocated if None);
        `out` may be such a view too."""
        t = self.torch
        assert x.is_cuda and x.dtype == t.float64 and x.shape[0] == self.S and x.shape[2] == self.channels
        frames = x.shape[1]
        assert x.stride(2) == 1 and x.stride(1) == self.channels and x.stride(0) % self.channels == 0 and x.stride(0) >= frames * self.channels
        in_stride = x.stride(0) // self.channels if self.S > 1 else frames
        cap = max(self.max_out_frames(frames), 1)
        if out is None:
            out = t.empty((self.S, cap, self.ochannels), dtype=t.float64, device=x.device)
        assert out.shape[0] == self.S and out.shape[2] == self.ochannels and out.shape[1] >= cap
        assert out.stride(2) == 1 and out.stride(1) == self.ochannels and out.stride(0) % self.ochannels == 0
        out_stride = out.stride(0) // self.ochannels if self.S > 1 else out.shape[1]
        f = self.L.dspamd_batch_run_strided(self.h, x.data_ptr(), in_stride, frames, out.data_ptr(), out_stride, self._stream())
        if f < 0:
            raise RuntimeError(f"dsp_amd: batch_run failed: {last_error()}")
        return out[:, :f, :]

    def drain(self, block, out=None):
        t = self.torch
        cap = max(self.max_out_frames(block), 1)
        if out is None:
            out = t.empty((self.S, cap, self.ochannels), dtype=t.float64, device="cuda")
        out_stride = out.stride(0) // self.ochannels if self.S > 1 else out.shape[1]
        f = self.L.dspamd_batch_drain(self.h, block, out.data_ptr(), out_stride, self._stream())
        if f == -1:
            return None
        if f < 0:
            raise RuntimeError(f"dsp_amd: batch_drain failed: {last_error()}")
        return out[:, :f, :]

    def reset(self):
        self.L.dspamd_batch_reset(self.h, self._stream())

    def process(self, x, block):
        """Whole streams incl. drain: x [S, N, C] cuda -> [S, M, C_out] cuda."""
        t = self.torch
        outs = []
        for p in range(0, x.shape[1], block):
            outs.append(self.run(x[:, p:p + block, :].contiguous()).clone())
        while True:
            o = self.drain(block)
            if o is None:
                break
            outs.append(o.clone())
        outs = [o for o in outs if o.shape[1]]
        return t.cat(outs, dim=1) if outs else t.zeros((self.S, 0, self.ochannels), dtype=t.float64, device=x.device)
