"""Python host over the C ABI, mirroring the reference's effects_chain life-cycle
(effects_chain.h:40-53): build -> run block by block -> drain -> destroy.

* ``EffectsChain``: one stream, numpy buffers on the host (PCIe each call) -- the
  compatibility path, same semantics as ``run_effects_chain`` / ``drain_effects_chain``.
* ``BatchChain``: S independent streams, torch CUDA tensors resident in HBM, launches
  on the current torch stream -- the throughput path.
"""
import numpy as np

from .lib import load_library, last_error


# DSPAMD_PCM_* (include/dsp_amd.h) and the torch dtype / elements per sample that carry each format
PCM_FORMATS = {"u8": 0, "s8": 1, "s16": 2, "s24": 3, "s32": 4, "s24_3": 5, "float": 6, "double": 7}
WIRE_DTYPES = {"u8": ("uint8", 1), "s8": ("int8", 1), "s16": ("int16", 1), "s24": ("int32", 1), "s32": ("int32", 1),
               "s24_3": ("uint8", 3), "float": ("float32", 1), "double": ("float64", 1)}


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

    def _check_out(self, out, dtype, row, cap):
        """an output slab [S, >= cap, row] of `dtype` on the device, rows contiguous (the kernels write through raw pointers: a
        wrong dtype or a strided view would put samples outside the tensor); returns its stream stride in frames"""
        if not (out.is_cuda and out.dtype == dtype and out.dim() == 3):
            raise ValueError(f"dsp_amd: output tensor must be a cuda tensor of {dtype} with 3 dimensions")
        if out.shape[0] != self.S or out.shape[2] != row or out.shape[1] < cap:
            raise ValueError(f"dsp_amd: output tensor must be [{self.S}, >= {cap}, {row}], got {tuple(out.shape)}")
        if out.stride(2) != 1 or out.stride(1) != row or (self.S > 1 and (out.stride(0) % row or out.stride(0) < out.shape[1] * row)):
            raise ValueError("dsp_amd: output tensor rows must be contiguous ([S, stride, row] buffer or a [:, :n, :] view of one)")
        return out.stride(0) // row if self.S > 1 else out.shape[1]

    def run(self, x, out=None):
        """x: [S, frames, C] cuda float64 -- contiguous, or a view x = buf[:, :frames, :] of a contiguous [S, stride, C] buffer
        (padded slabs: see dspamd_batch_run_strided).  Returns a view [S, oframes, C_out] of `out` (allocated if None);
        `out` may be such a view too."""
        t = self.torch
        assert x.is_cuda and x.dtype == t.float64 and x.shape[0] == self.S and x.shape[2] == self.channels
        frames = x.shape[1]
        assert x.stride(2) == 1 and x.stride(1) == self.channels
        assert self.S == 1 or (x.stride(0) % self.channels == 0 and x.stride(0) >= frames * self.channels)
        in_stride = x.stride(0) // self.channels if self.S > 1 else frames
        cap = max(self.max_out_frames(frames), 1)
        if out is None:
            out = t.empty((self.S, cap, self.ochannels), dtype=t.float64, device=x.device)
        out_stride = self._check_out(out, t.float64, self.ochannels, cap)
        f = self.L.dspamd_batch_run_strided(self.h, x.data_ptr(), in_stride, frames, out.data_ptr(), out_stride, self._stream())
        if f < 0:
            raise RuntimeError(f"dsp_amd: batch_run failed: {last_error()}")
        return out[:, :f, :]

    def drain(self, block, out=None):
        t = self.torch
        cap = max(self.max_out_frames(block), 1)
        if out is None:
            out = t.empty((self.S, cap, self.ochannels), dtype=t.float64, device="cuda")
        out_stride = self._check_out(out, t.float64, self.ochannels, cap)
        f = self.L.dspamd_batch_drain(self.h, block, out.data_ptr(), out_stride, self._stream())
        if f == -1:
            return None
        if f < 0:
            raise RuntimeError(f"dsp_amd: batch_drain failed: {last_error()}")
        return out[:, :f, :]

    def reset(self):
        self.L.dspamd_batch_reset(self.h, self._stream())

    # ---- wire format to wire format (the file -> file path: read_buf_<fmt>, the chain, dither / clip / write_buf_<fmt>) ----
    def _wire_out(self, fmt, cap, device):
        t = self.torch
        dt, mult = WIRE_DTYPES[fmt]
        return t.empty((self.S, cap, self.ochannels * mult), dtype=getattr(t, dt), device=device)

    def run_wire(self, x, in_fmt, out_fmt, dither_prec=0, stats=None, out=None):
        """x: [S, frames, C] cuda tensor of the wire dtype of in_fmt (WIRE_DTYPES; s24_3: uint8 [S, frames, 3 C]), contiguous or a
        view of a padded [S, stride, C] buffer.  Returns a view [S, oframes, C_out] of `out` in the dtype of out_fmt.  The batch
        keeps the position in the dither sequences; stats: optional zero-initialised [S, 2] float64 (clip count bits, peak)."""
        t = self.torch
        dt, mult = WIRE_DTYPES[in_fmt]
        assert x.is_cuda and x.dtype == getattr(t, dt) and x.shape[0] == self.S and x.shape[2] == self.channels * mult
        frames, row = x.shape[1], self.channels * mult
        assert x.stride(2) == 1 and x.stride(1) == row and (self.S == 1 or (x.stride(0) % row == 0 and x.stride(0) >= frames * row))
        in_stride = x.stride(0) // row if self.S > 1 else frames
        cap = max(self.max_out_frames(frames), 1)
        if out is None:
            out = self._wire_out(out_fmt, cap, x.device)
        odt, omult = WIRE_DTYPES[out_fmt]
        out_stride = self._check_out(out, getattr(t, odt), self.ochannels * omult, cap)
        f = self.L.dspamd_batch_run_wire(self.h, PCM_FORMATS[in_fmt], x.data_ptr(), in_stride, frames, PCM_FORMATS[out_fmt], out.data_ptr(), out_stride,
                                         dither_prec, stats.data_ptr() if stats is not None else None, self._stream())
        if f < 0:
            raise RuntimeError(f"dsp_amd: batch_run_wire failed: {last_error()}")
        return out[:, :f, :]

    def drain_wire(self, block, out_fmt, dither_prec=0, stats=None, out=None):
        cap = max(self.max_out_frames(block), 1)
        if out is None:
            out = self._wire_out(out_fmt, cap, "cuda")
        odt, omult = WIRE_DTYPES[out_fmt]
        out_stride = self._check_out(out, getattr(self.torch, odt), self.ochannels * omult, cap)
        f = self.L.dspamd_batch_drain_wire(self.h, block, PCM_FORMATS[out_fmt], out.data_ptr(), out_stride, dither_prec,
                                           stats.data_ptr() if stats is not None else None, self._stream())
        if f == -1:
            return None
        if f < 0:
            raise RuntimeError(f"dsp_amd: batch_drain_wire failed: {last_error()}")
        return out[:, :f, :]

    def wire_fused(self):
        """what the last run_wire / drain_wire did: bit 0 = input converted by the first kernel, bit 1 = sink applied by the last"""
        return self.L.dspamd_batch_wire_fused(self.h)

    def process_wire(self, x, block, in_fmt, out_fmt, dither_prec=0, stats=None):
        """Whole streams incl. drain, wire format to wire format."""
        t = self.torch
        outs = []
        for p in range(0, x.shape[1], block):
            outs.append(self.run_wire(x[:, p:p + block, :].contiguous(), in_fmt, out_fmt, dither_prec, stats).clone())
        while True:
            o = self.drain_wire(block, out_fmt, dither_prec, stats)
            if o is None:
                break
            outs.append(o.clone())
        outs = [o for o in outs if o.shape[1]]
        return t.cat(outs, dim=1) if outs else self._wire_out(out_fmt, 1, x.device)[:, :0, :]

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
