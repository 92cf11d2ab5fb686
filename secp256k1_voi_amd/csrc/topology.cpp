// topology.cpp — which host CPUs and which memory are close to a device.
//
// The reference runs on the caller's goroutine and has no notion of placement.  A node with eight MI355X has two CPU
// sockets; a device's DMA engine reaches the memory of the OTHER socket over the socket link (a fraction of the PCIe rate
// when four devices pull across it at once), and a member thread of a device group that floats between the sockets pays
// that on every submit.  This file answers, from sysfs alone (no libnuma in the image):
//   device -> PCI bus id (s2k_device_pci_bus_id, engine.hip: the only HIP call) -> <sysfs>/bus/pci/devices/<id>/numa_node
//   node   -> <sysfs>/devices/system/node/node<N>/cpulist
// and binds the calling thread (sched_setaffinity) or a range of pages (mbind through the raw system call) accordingly.
// Everything degrades to a no-op on a single-node machine, inside a container that hides the node files, or when the
// kernel refuses.  S2K_SYSFS_ROOT replaces "/sys" (tests/test_topology_cpu.py builds a fake tree); pure host code.
#include <dirent.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/secp256k1_voi_amd.h"

namespace {

const char* sysfs_root(const char* given) {
  if (given && *given) return given;
  const char* e = getenv("S2K_SYSFS_ROOT");
  return e && *e ? e : "/sys";
}

bool read_small_file(const std::string& path, char* buf, size_t cap) {
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return false;
  const size_t n = fread(buf, 1, cap - 1, f);
  fclose(f);
  buf[n] = 0;
  return n > 0;
}

// "0-15,32-47\n" -> cpu numbers; false on anything else
bool parse_cpulist(const char* s, std::vector<int>& out) {
  out.clear();
  const char* p = s;
  while (*p && *p != '\n') {
    char* end = nullptr;
    errno = 0;
    const long a = strtol(p, &end, 10);
    if (end == p || errno || a < 0 || a > 1 << 20) return false;
    long b = a;
    p = end;
    if (*p == '-') {
      ++p;
      b = strtol(p, &end, 10);
      if (end == p || errno || b < a || b > 1 << 20) return false;
      p = end;
    }
    for (long c = a; c <= b; ++c) out.push_back((int)c);
    if (*p == ',') ++p;
    else if (*p && *p != '\n') return false;
  }
  return true;
}

}  // namespace

extern "C" {

// number of NUMA nodes the kernel shows (node<N> directories); 0 when the directory is missing (then nothing is bound)
int s2k_topology_node_count(const char* root) {
  const std::string dir = std::string(sysfs_root(root)) + "/devices/system/node";
  DIR* d = opendir(dir.c_str());
  if (!d) return 0;
  int count = 0;
  while (dirent* e = readdir(d)) {
    if (strncmp(e->d_name, "node", 4) != 0) continue;
    char* end = nullptr;
    (void)strtol(e->d_name + 4, &end, 10);
    if (end != e->d_name + 4 && *end == 0) ++count;
  }
  closedir(d);
  return count;
}

// NUMA node of a PCI function ("0000:05:00.0"; case as sysfs has it: lower), -1 when unknown (file missing, or the
// kernel's own -1 on single-node machines and in most virtual machines)
int s2k_topology_numa_node_of_pci(const char* root, const char* bus_id) {
  if (!bus_id || !*bus_id) return -1;
  std::string id(bus_id);
  for (char& c : id) c = (char)tolower((unsigned char)c);
  char buf[64];
  if (!read_small_file(std::string(sysfs_root(root)) + "/bus/pci/devices/" + id + "/numa_node", buf, sizeof buf)) return -1;
  char* end = nullptr;
  const long v = strtol(buf, &end, 10);
  return end == buf || v < 0 || v > 4096 ? -1 : (int)v;
}

// the CPUs of a node into cpus[0 .. cap); returns how many the node has (may exceed cap), 0 when unknown
int s2k_topology_node_cpus(const char* root, int node, int* cpus, size_t cap) {
  if (node < 0) return 0;
  char buf[4096];
  if (!read_small_file(std::string(sysfs_root(root)) + "/devices/system/node/node" + std::to_string(node) + "/cpulist", buf, sizeof buf)) return 0;
  std::vector<int> v;
  if (!parse_cpulist(buf, v)) return 0;
  for (size_t i = 0; i < v.size() && i < cap; ++i) cpus[i] = v[i];
  return (int)v.size();
}

int s2k_device_numa_node(int device) {
  char id[64] = {0};
  if (s2k_device_pci_bus_id(device, id, sizeof id) != S2K_OK) return -1;
  return s2k_topology_numa_node_of_pci(nullptr, id);
}

// Restricts the CALLING thread to the CPUs of `node` that it is allowed to run on now.  Returns the number of CPUs it is
// bound to, 0 when nothing was changed: unknown node, one node only, no CPU of the node allowed (cgroup cpuset), or the
// kernel refused.  Never widens the thread's mask.
int s2k_bind_thread_to_node(int node) {
  if (node < 0 || s2k_topology_node_count(nullptr) < 2) return 0;
  std::vector<int> cpus(4096);
  const int n = s2k_topology_node_cpus(nullptr, node, cpus.data(), cpus.size());
  if (n <= 0) return 0;
  cpu_set_t now, want;
  CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof now, &now) != 0) return 0;
  int bound = 0;
  for (int i = 0; i < n && i < (int)cpus.size(); ++i)
    if (cpus[i] < CPU_SETSIZE && CPU_ISSET(cpus[i], &now)) {
      CPU_SET(cpus[i], &want);
      ++bound;
    }
  if (bound == 0) return 0;
  if (sched_setaffinity(0, sizeof want, &want) != 0) return 0;
  return bound;
}

// Asks the kernel to place the pages of [p, p + bytes) on `node` (MPOL_PREFERRED: falls back to other nodes instead of
// failing when the node is full).  p and bytes are rounded inwards to whole pages.  Returns 0 on success or when there is
// nothing to do (unknown node, one node), -1 when the kernel refused (the pages then follow first touch).
int s2k_topology_prefer_node(void* p, size_t bytes, int node) {
  if (node < 0 || node >= 1024 || s2k_topology_node_count(nullptr) < 2) return 0;
  const size_t page = (size_t)sysconf(_SC_PAGESIZE);
  uintptr_t lo = ((uintptr_t)p + page - 1) & ~(uintptr_t)(page - 1), hi = ((uintptr_t)p + bytes) & ~(uintptr_t)(page - 1);
  if (hi <= lo) return 0;
  unsigned long mask[1024 / (8 * sizeof(unsigned long))] = {0};
  mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
#ifdef SYS_mbind
  const int MPOL_PREFERRED_ = 1;
  return syscall(SYS_mbind, (void*)lo, (unsigned long)(hi - lo), MPOL_PREFERRED_, mask, (unsigned long)1024, 0u) == 0 ? 0 : -1;
#else
  return -1;
#endif
}

}  // extern "C"
