"""csrc/topology.cpp on a fake sysfs tree: which NUMA node a PCI function hangs off, which CPUs a node
has, how a thread is bound - the parts of the device group's placement that can be checked without an eight-GPU node.
Runs without a GPU; the library has to be built."""
import ctypes as C
import os

import pytest


@pytest.fixture(scope="module")
def lib():
    import secp256k1_voi_amd as S
    if not os.path.exists(S.LIB_PATH):
        pytest.skip("library not built")
    return S.load_library()


def make_tree(root, nodes, devices):
    """nodes: {node: cpulist string}; devices: {bus id: numa_node file content}"""
    for node, cpus in nodes.items():
        d = root / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    (root / "devices" / "system" / "node" / "possible").write_text("0-1\n")        # a file that is not a node directory
    (root / "devices" / "system" / "node" / "nodeX").mkdir(exist_ok=True)          # nor is this
    for bus, val in devices.items():
        d = root / "bus" / "pci" / "devices" / bus
        d.mkdir(parents=True)
        if val is not None:
            (d / "numa_node").write_text(val + "\n")


def test_parsers_on_a_two_socket_tree(lib, tmp_path):
    make_tree(tmp_path, {0: "0-47,96-143", 1: "48-95,144-191"},
              {"0000:05:00.0": "0", "0000:c5:00.0": "1", "0000:e5:00.0": "-1", "0000:f5:00.0": None, "0000:a5:00.0": "junk"})
    root = str(tmp_path).encode()
    assert lib.s2k_topology_node_count(root) == 2
    assert lib.s2k_topology_numa_node_of_pci(root, b"0000:05:00.0") == 0
    assert lib.s2k_topology_numa_node_of_pci(root, b"0000:C5:00.0") == 1          # hipDeviceGetPCIBusId prints upper case
    assert lib.s2k_topology_numa_node_of_pci(root, b"0000:e5:00.0") == -1         # the kernel's own "unknown"
    assert lib.s2k_topology_numa_node_of_pci(root, b"0000:f5:00.0") == -1         # no such file
    assert lib.s2k_topology_numa_node_of_pci(root, b"0000:a5:00.0") == -1         # garbage
    assert lib.s2k_topology_numa_node_of_pci(root, b"0000:00:00.0") == -1         # no such device
    assert lib.s2k_topology_numa_node_of_pci(root, b"") == -1
    cpus = (C.c_int * 256)()
    assert lib.s2k_topology_node_cpus(root, 0, cpus, 256) == 96
    assert list(cpus[:96]) == list(range(0, 48)) + list(range(96, 144))
    assert lib.s2k_topology_node_cpus(root, 1, cpus, 4) == 96 and list(cpus[:4]) == [48, 49, 50, 51]   # count even when the buffer is short
    assert lib.s2k_topology_node_cpus(root, 2, cpus, 256) == 0
    assert lib.s2k_topology_node_cpus(root, -1, cpus, 256) == 0


def test_cpulist_forms(lib, tmp_path):
    for i, (text, want) in enumerate([("3", [3]), ("0,2,4", [0, 2, 4]), ("0-1,7,9-10", [0, 1, 7, 9, 10]), ("", None), ("5-3", None),
                                      ("a-b", None), ("1,,2", None)]):
        root = tmp_path / f"t{i}"
        make_tree(root, {0: text}, {})
        cpus = (C.c_int * 16)()
        n = lib.s2k_topology_node_cpus(str(root).encode(), 0, cpus, 16)
        assert (list(cpus[:n]) if n else None) == want, text


def test_binding_in_a_child_process(tmp_path):
    """s2k_bind_thread_to_node through S2K_SYSFS_ROOT, in a child process (the binding sticks to the thread): node 1 of the
    fake tree holds the second half of the CPUs this process may use; one node, an unknown node, or a node without any
    allowed CPU change nothing; the mask is never widened."""
    import subprocess
    import sys
    import textwrap
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("needs two CPUs")
    half = len(allowed) // 2
    fmt = lambda xs: ",".join(str(x) for x in xs)
    make_tree(tmp_path / "two", {0: fmt(allowed[:half]), 1: fmt(allowed[half:]), 2: "100000"}, {})
    make_tree(tmp_path / "one", {0: fmt(allowed)}, {})
    code = textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
        import secp256k1_voi_amd as S
        lib = S.load_library()
        before = sorted(os.sched_getaffinity(0))
        os.environ["S2K_SYSFS_ROOT"] = {str(tmp_path / "one")!r}
        assert lib.s2k_bind_thread_to_node(0) == 0 and sorted(os.sched_getaffinity(0)) == before          # one node: nothing to choose
        os.environ["S2K_SYSFS_ROOT"] = {str(tmp_path / "two")!r}
        assert lib.s2k_bind_thread_to_node(-1) == 0 and lib.s2k_bind_thread_to_node(7) == 0
        assert lib.s2k_bind_thread_to_node(2) == 0 and sorted(os.sched_getaffinity(0)) == before          # a node whose CPUs are not ours
        n = lib.s2k_bind_thread_to_node(1)
        assert n == {len(allowed) - half} and sorted(os.sched_getaffinity(0)) == {allowed[half:]!r}, (n, sorted(os.sched_getaffinity(0)))
        assert lib.s2k_bind_thread_to_node(0) == 0 and sorted(os.sched_getaffinity(0)) == {allowed[half:]!r}   # never widened: node 0's CPUs are gone
        buf = bytearray(1 << 16)
        import ctypes
        p = ctypes.addressof((ctypes.c_char * len(buf)).from_buffer(buf))
        assert lib.s2k_topology_prefer_node(p, len(buf), -1) == 0                                       # unknown node: nothing to do
        assert lib.s2k_topology_prefer_node(p, len(buf), 1) in (0, -1)                                  # the real kernel may refuse the fake node
        print("ok")
        """)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stdout + p.stderr
