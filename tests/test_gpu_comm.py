"""The C-level multi-GPU exchange (include/unina_mi355.h "multi-GPU", csrc/comm.hip): what a C++ node with several MI355X calls
instead of gather.py's torch.distributed loop. On the one GPU of the test box: a communicator of world size 1 over RCCL
(loaded by the library on first use), the all-gather of detection slots on a stream of its own behind the frames' event,
and the result byte-equal to the serial unina_infer of the same frames -- the same check tests/test_gpu_rccl.py makes for
the Python loop. World size 2 needs two GPUs: the driver's scaling run only."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_c_level_gather_of_detection_slots_world_size_1(pkg, sd7):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from unina_yolo_dla_amd.engine import Engine, MAX_DETECTIONS
    e = Engine.from_state_dict(sd7)
    L = e.L
    try:
        ident = (C.c_char * 128)()
        rc = L.unina_comm_unique_id(ident)
        assert rc == 0, L.unina_comm_last_error()
        comm = C.c_void_p()
        rc = L.unina_comm_init(C.byref(comm), ident, 0, 1, 0)
        assert rc == 0 and comm.value, L.unina_comm_last_error()
        assert L.unina_comm_rank(comm) == 0 and L.unina_comm_world(comm) == 1
        k, words = 8, 8 + 8 * MAX_DETECTIONS
        local = torch.zeros((k, words), dtype=torch.int32, device="cuda")
        gathered = torch.full((1, k, words), -1, dtype=torch.int32, device="cuda")
        frames = [torch.from_numpy(pkg.rng.frame(1234 + i, 640, 640)).cuda() for i in range(k)]
        infer_stream, comm_stream = torch.cuda.Stream(), torch.cuda.Stream()
        with torch.cuda.stream(infer_stream):
            for i in range(k):
                e.infer_async(frames[i], 0.5, 0.45, 0.1, out=local[i])
            done = torch.cuda.Event()
            done.record(infer_stream)
        comm_stream.wait_event(done)                      # the gather runs behind exactly the frames it carries
        rc = L.unina_comm_all_gather(comm, local.data_ptr(), gathered.data_ptr(), local.numel() * 4, comm_stream.cuda_stream)
        assert rc == 0, L.unina_comm_last_error()
        comm_stream.synchronize()
        got = gathered.cpu().numpy()[0]
        for i in range(k):
            want = e.infer(frames[i], 0.5, 0.45, 0.1)
            assert e.unpack(torch.from_numpy(got[i])).tobytes() == want.tobytes(), i
        # argument errors are reported, not fatal
        assert L.unina_comm_all_gather(comm, None, gathered.data_ptr(), 16, None) != 0
        assert b"bad arguments" in L.unina_comm_last_error()
        L.unina_comm_destroy(comm)
    finally:
        e.close()
