"""Host-side placement of a rank: which CPUs its threads (the hypothesis lanes, RCCL's proxies) may run on.

One process per GPU; a rank's lanes wait for ITS device and solve small dense problems between the waits, so they belong on the
cores of the NUMA node that device hangs off - on a two-socket MI355X host the wrong socket puts every pinned-memory copy and
every doorbell write across the inter-socket link.  ``bind_rank_to_device_numa`` works from sysfs alone (the KFD topology and the
PCI device's ``numa_node`` / ``local_cpulist``), so it can - and must - run BEFORE the process first touches the GPU: threads the
HIP runtime starts afterwards inherit the mask.  Nothing here calls into HIP.

The reference pins nothing (experiments/material_sync_train.py:23 selects one device and runs one process); this is part of the
N > 1 path the north star adds.
"""
import os


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def _visible_order(ngpu, env):
    """Indices into the KFD GPU list in the order HIP numbers the devices (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES as plain integer lists; anything else - UUIDs - gives None: unknown)."""
    order = list(range(ngpu))
    for name in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = env.get(name)
        if val is None or val.strip() == "":
            continue
        try:
            pick = [int(x) for x in val.split(",") if x.strip() != ""]
        except ValueError:
            return None
        if any(i < 0 or i >= len(order) for i in pick):
            return None
        order = [order[i] for i in pick]
    return order


def kfd_gpus(root="/"):
    """PCI addresses of the GPU nodes of the KFD topology, in node order (the order the HIP runtime enumerates them in)."""
    base = os.path.join(root, "sys/class/kfd/kfd/topology/nodes")
    out = []
    try:
        nodes = sorted((d for d in os.listdir(base) if d.isdigit()), key=int)
    except OSError:
        return out
    for d in nodes:
        props = {}
        try:
            with open(os.path.join(base, d, "properties")) as f:
                for line in f:
                    kv = line.split()
                    if len(kv) == 2:
                        props[kv[0]] = kv[1]
        except OSError:
            continue
        try:
            if int(props.get("simd_count", "0")) <= 0:
                continue  # a CPU node
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
        except (KeyError, ValueError):
            continue
        bus, devfn = (loc >> 8) & 0xFF, loc & 0xFF
        out.append(f"{dom:04x}:{bus:02x}:{(devfn >> 3) & 0x1F:02x}.{devfn & 7}")
    return out


def device_numa(local_rank, root="/", env=None):
    """(numa_node, cpus, pci_address) of HIP device ``local_rank`` from sysfs; numa_node None / cpus [] when unknown."""
    env = os.environ if env is None else env
    gpus = kfd_gpus(root)
    order = _visible_order(len(gpus), env)
    if not gpus or order is None or local_rank < 0 or local_rank >= len(order):
        return None, [], None
    addr = gpus[order[local_rank]]
    dev = os.path.join(root, "sys/bus/pci/devices", addr)
    node, cpus = None, []
    try:
        with open(os.path.join(dev, "numa_node")) as f:
            node = int(f.read().strip())
    except (OSError, ValueError):
        pass
    try:
        with open(os.path.join(dev, "local_cpulist")) as f:
            cpus = parse_cpulist(f.read())
    except (OSError, ValueError):
        pass
    if node is not None and node < 0:
        node = None  # (a single-node host, or a guest that hides the topology)
    return node, cpus, addr


def bind_rank_to_device_numa(local_rank, local_world=1, min_cpus=4, root="/", env=None, apply=True):
    """Restrict the calling process to the CPUs local to HIP device ``local_rank`` - its share of them when several ranks of
    the node hang off the same NUMA node (contiguous slices in rank order) - and return a record for the benchmark line:
    {"numa_node", "pci", "cpus" (count), "cpu_list" (compact), "bound": bool, "why"}.  Call it before anything initialises the GPU.
    Never raises: an unknown topology leaves the affinity as it is and says so."""
    rec = {"numa_node": None, "pci": None, "cpus": None, "cpu_list": None, "bound": False, "why": ""}
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        rec["why"] = "no sched_getaffinity on this platform"
        return rec
    rec["cpus"] = len(allowed)
    node, cpus, addr = device_numa(local_rank, root, env)
    rec["numa_node"], rec["pci"] = node, addr
    if addr is None:
        rec["why"] = "KFD topology not readable (or device selection by UUID): affinity left as inherited"
        return rec
    local = [c for c in cpus if c in set(allowed)]
    if node is None or not local or len(local) == len(allowed):
        rec["why"] = "the device reports no NUMA node of its own (single-node host or hidden topology): nothing to bind to"
        return rec
    # the ranks of this node that share the NUMA node split its CPUs in rank order
    sharing = [r for r in range(max(1, local_world)) if device_numa(r, root, env)[0] == node]
    if local_rank in sharing and len(sharing) > 1:
        per = len(local) // len(sharing)
        if per >= min_cpus:
            i = sharing.index(local_rank)
            local = local[i * per:(i + 1) * per]
    if len(local) < min_cpus:
        rec["why"] = f"only {len(local)} local CPUs allowed: affinity left as inherited"
        return rec
    if apply:
        try:
            os.sched_setaffinity(0, local)
        except OSError as ex:
            rec["why"] = f"sched_setaffinity refused: {ex}"
            return rec
    rec.update(bound=bool(apply), cpus=len(local), cpu_list=compact(local),
               why=f"CPUs of NUMA node {node} (device {addr})" + (f", share {sharing.index(local_rank) + 1} of {len(sharing)}" if len(sharing) > 1 and local_rank in sharing else ""))
    return rec


def compact(cpus):
    """[0, 1, 2, 3, 8] -> '0-3,8'"""
    cpus = sorted(cpus)
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)
