"""
Gate circuits as ONE device launch: hipGraph capture of a sequence of gate calls (SURVEY 8f row 4 "graph capture";
the reference's example of a circuit is uint_min, nufhe/operators_integer.py:64-95 -- 2 x itemsize dependent gates, each
of which the reference issues as ~10 kernel launches through Reikna).

Every launch of this package goes to one HIP stream per DeviceThread and nothing on the gate path synchronises or
allocates through the library after the first call of a given size, so a circuit can be recorded once and replayed with
new inputs in place:

    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):                      # capture needs a stream of its own (not the default stream)
        ctx = nufhe.Context(rng=..., thread=DeviceThread(0))
        secret, cloud = ctx.make_key_pair()
        vm = ctx.make_virtual_machine(cloud)
        a, b = ctx.encrypt(secret, bits_a), ctx.encrypt(secret, bits_b)   # static input buffers
        circuit = GateGraph(ctx.thread)
        out = circuit.capture(lambda: vm.gate_mux(vm.gate_nand(a, b), a, b))      # records (nothing runs yet)
        circuit.replay()                                 # one launch for the whole circuit; `out` holds the result
        a[...] = ctx.encrypt(secret, other_bits)         # new inputs, written IN PLACE ...
        circuit.replay()                                 # ... and the same graph again

What it buys is the host side: the Python / ctypes / allocator work per gate (~60-100 us) and the launch gaps between
the 3-5 kernels of every gate disappear; the device work (500 dependent blind-rotation steps per gate) is unchanged --
measured (profiles/r04_graph_capture.json): uint_min on 4 x 16 bits 85.4 ms eager, 85.3 ms replayed; a 16-gate NAND chain
on 4 bits 75.4 / 75.6 ms.  The gates are so long (4.7 ms each) that the stream never runs dry in eager mode either; the
graph is for hosts that must not spend a core on issuing gates, not a latency win.

The captured graph holds raw pointers into the library's scratch buffers (extracted samples, keyswitch accumulators),
which grow on demand.  A GateGraph therefore PINS the context's scratch from its first capture until ``close()`` (or
garbage collection): a later, larger eager gate on the same DeviceThread allocates a new buffer and the old one stays
alive for the graph (``nufhe_ctx_pin_scratch``) -- replaying an old graph after larger gates have run is safe.
``VirtualMachine.gate_batch`` is capturable like the single gates (its job tables travel inside kernel arguments).
"""

import torch

from . import _lib


class GateGraph:

    def __init__(self, thread):
        self.thread = thread
        self.graph = None
        self.outputs = None
        self._pinned = False
        if thread._torch_stream.cuda_stream == 0:
            raise ValueError(
                "the default stream cannot be captured: create the DeviceThread / Context inside "
                "`with torch.cuda.stream(torch.cuda.Stream()):` and use it there")

    def capture(self, circuit, warmup=1):
        """Runs ``circuit()`` ``warmup`` times eagerly (the library sizes its scratch buffers and per-key layouts on
        first use), then RECORDS one more run into a graph -- a recording executes nothing; ``replay()`` does.  Returns
        what ``circuit`` returned during the recording: those ciphertexts are the graph's OUTPUT BUFFERS, written by
        every ``replay``; the ciphertexts the circuit
        read are its INPUT BUFFERS -- refresh them in place (``ct[...] = other``)."""
        stream = self.thread._torch_stream
        if not self._pinned:
            # (straight through the binding: pinning touches no device memory, so it is not subject to the guard that
            # ties calls to the DeviceThread's stream -- capture() may be called from outside that stream's context)
            _lib.check(_lib.lib().nufhe_ctx_pin_scratch(self.thread.handle, 1))
            self._pinned = True
        with torch.cuda.stream(stream):
            for _ in range(max(1, int(warmup))):
                circuit()
            self.thread.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                self.outputs = circuit()
        self.graph = graph
        return self.outputs

    def replay(self):
        if self.graph is None:
            raise RuntimeError("nothing captured yet")
        with torch.cuda.stream(self.thread._torch_stream):
            self.graph.replay()
        return self.outputs

    def close(self):
        """Drops the graph and releases its pin on the context's scratch buffers."""
        self.graph = None
        if self._pinned:
            self._pinned = False
            handle = getattr(self.thread, 'handle', None)
            if handle is not None and handle.value is not None and not getattr(self.thread, '_released', False):
                _lib.check(_lib.lib().nufhe_ctx_pin_scratch(handle, -1))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
