"""Writer / reader of the recorded-sequence file tools/ll_node.cpp replays (stands in for a rosbag of /laser_points_<i>):
"LLSEQ001", int32 n_messages, then per message { int32 lidar_index, float64 stamp, int32 n_points, n x 4 float32 }."""
import struct

import numpy as np


def write_sequence(path, messages):
    """messages: iterable of (lidar_index, stamp, xyzi (n, 4) float32)"""
    messages = list(messages)
    with open(path, "wb") as f:
        f.write(b"LLSEQ001" + struct.pack("<i", len(messages)))
        for lidar, stamp, xyzi in messages:
            a = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
            f.write(struct.pack("<idi", int(lidar), float(stamp), a.shape[0]))
            f.write(a.tobytes())


def read_sequence(path):
    out = []
    with open(path, "rb") as f:
        assert f.read(8) == b"LLSEQ001"
        (n,) = struct.unpack("<i", f.read(4))
        for _ in range(n):
            lidar, stamp, k = struct.unpack("<idi", f.read(16))
            out.append((lidar, stamp, np.frombuffer(f.read(16 * k), np.float32).reshape(k, 4).copy()))
    return out


def parse_log(path):
    """ll_node's log -> (pub rows [(n_full, n_surf, n_corner, h_full, h_surf, h_corner)], reg rows [dict])"""
    pubs, regs = [], []
    for line in open(path):
        w = line.split()
        if w[0] == "PUB":
            pubs.append((int(w[1]), int(w[2]), int(w[3]), int(w[4], 16), int(w[5], 16), int(w[6], 16)))
        elif w[0] == "REG":
            regs.append(dict(frame=int(w[1]), res=int(w[2]), pose=np.array([float(v) for v in w[3:10]]), n_corner=int(w[10]), n_surf=int(w[11]),
                             map_corner=int(w[12]), map_surf=int(w[13]), icp_iterations=int(w[14])))
    return pubs, regs


def cloud_hash(a: np.ndarray) -> int:
    """position-weighted sum of the cloud's 32-bit words, mod 2^64 (what ll_node logs for each published cloud)"""
    w = np.ascontiguousarray(a, np.float32).view(np.uint32).ravel().astype(np.uint64)
    with np.errstate(over="ignore"):
        return int((w * (2 * np.arange(len(w), dtype=np.uint64) + 1)).sum(dtype=np.uint64))
