"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's instance-type catalog arithmetic.

Follows, line by line (paths relative to the reference checkout):
  pkg/providers/instancetype/types.go:51-478      NewInstanceType, computeRequirements,
                                                  computeCapacity, memory, ephemeralStorage, pods,
                                                  ENILimitedPods, kubeReservedResources,
                                                  evictionThreshold, computeEvictionSignal
  pkg/providers/instancetype/instancetype.go:264-315  createOfferings
  pkg/providers/amifamily/resolver.go:37-41,93-111    DefaultEBS (20Gi), default FeatureFlags
  pkg/providers/amifamily/bottlerocket.go:105-144, windows.go:72-93, al2023.go:69-79
  pkg/providers/pricing/pricing.go:132-157            OnDemandPrice / SpotPrice lookups

Pinned by the 776 known-answer rows of website/content/en/preview/reference/instance-types.md
(tests/golden/instance_types_doc.json) and the exact values quoted in
pkg/providers/instancetype/suite_test.go (tests/test_catalog_oracle.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
All quantities are int milli-units, as in include/karpgpu.h.
"""
import math
import re

MI = 1024 * 1024
GI = 1024 * MI

RES_CPU, RES_MEM, RES_EPH, RES_PODS = "cpu", "memory", "ephemeral-storage", "pods"
RES_POD_ENI = "vpc.amazonaws.com/pod-eni"
RES_NVIDIA, RES_AMD, RES_NEURON = "nvidia.com/gpu", "amd.com/gpu", "aws.amazon.com/neuron"
RES_GAUDI, RES_EFA = "habana.ai/gaudi", "vpc.amazonaws.com/efa"
RES_PRIVATE_IPV4 = "vpc.amazonaws.com/PrivateIPv4Address"

G = "karpenter.k8s.aws"
L_INSTANCE_TYPE, L_ARCH, L_OS = "node.kubernetes.io/instance-type", "kubernetes.io/arch", "kubernetes.io/os"
L_ZONE, L_REGION, L_WINDOWS_BUILD = "topology.kubernetes.io/zone", "topology.kubernetes.io/region", "node.kubernetes.io/windows-build"
L_CAPACITY_TYPE, L_ZONE_ID = "karpenter.sh/capacity-type", "topology.k8s.aws/zone-id"
L_NODEPOOL, L_HOSTNAME = "karpenter.sh/nodepool", "kubernetes.io/hostname"

# karpv1.WellKnownLabels after pkg/apis/v1/labels.go:29-53 has run
WELL_KNOWN_LABELS = [
    L_NODEPOOL, L_ZONE, L_REGION, L_INSTANCE_TYPE, L_ARCH, L_OS, L_CAPACITY_TYPE, L_WINDOWS_BUILD,
    G + "/instance-hypervisor", G + "/instance-encryption-in-transit-supported", G + "/instance-category",
    G + "/instance-family", G + "/instance-generation", G + "/instance-size", G + "/instance-local-nvme",
    G + "/instance-cpu", G + "/instance-cpu-manufacturer", G + "/instance-memory", G + "/instance-ebs-bandwidth",
    G + "/instance-network-bandwidth", G + "/instance-gpu-name", G + "/instance-gpu-manufacturer",
    G + "/instance-gpu-count", G + "/instance-gpu-memory", G + "/instance-accelerator-name",
    G + "/instance-accelerator-manufacturer", G + "/instance-accelerator-count", L_ZONE_ID,
]

_SUFFIX = {"Ki": 1024, "Mi": 1024 ** 2, "Gi": 1024 ** 3, "Ti": 1024 ** 4, "Pi": 1024 ** 5, "Ei": 1024 ** 6,
           "k": 10 ** 3, "M": 10 ** 6, "G": 10 ** 9, "T": 10 ** 12, "P": 10 ** 15, "E": 10 ** 18, "": 1}


def parse_quantity(s):
    """resource.MustParse -> exact int milli-units (raises if finer than milli)."""
    from fractions import Fraction
    s = str(s).strip()
    m = re.fullmatch(r"([+-]?[0-9.]+(?:[eE][+-]?[0-9]+)?)(m|Ki|Mi|Gi|Ti|Pi|Ei|k|M|G|T|P|E|)", s)
    if not m:
        raise ValueError("bad quantity %r" % s)
    num, suf = m.group(1), m.group(2)
    if "e" in num.lower():
        mant, exp = re.split("[eE]", num)
        val = Fraction(mant) * Fraction(10) ** int(exp)
    else:
        val = Fraction(num)
    milli = val * (1 if suf == "m" else 1000 * _SUFFIX[suf])
    # resource.Quantity rounds up to the nearest milli on parse of finer values
    if milli.denominator != 1:
        milli = Fraction(math.ceil(milli))
    return int(milli)


class AMIFamily:
    def __init__(self, name):
        self.name = name
        self.windows = name in ("Windows2019", "Windows2022")
        self.custom = name == "Custom"
        if name == "Bottlerocket":  # bottlerocket.go:137-144
            self.flags = dict(eni_mem=False, pods_per_core=False, eviction_soft=False, eni_pods=True)
            self.default_bdm = [("/dev/xvda", 4 * GI * 1000), ("/dev/xvdb", 20 * GI * 1000)]
            self.ephemeral_device = "/dev/xvdb"
        elif self.windows:  # windows.go:72-93
            self.flags = dict(eni_mem=False, pods_per_core=True, eviction_soft=True, eni_pods=False)
            self.default_bdm = [("/dev/sda1", 50 * GI * 1000)]
            self.ephemeral_device = "/dev/sda1"
            self.build = "10.0.17763" if name == "Windows2019" else "10.0.20348"
        else:  # resolver.go:100-107 DefaultFamily
            self.flags = dict(eni_mem=True, pods_per_core=True, eviction_soft=True, eni_pods=True)
            if self.custom:
                self.default_bdm, self.ephemeral_device = [], None
            else:
                dev = "/dev/sda1" if name == "Ubuntu" else "/dev/xvda"
                self.default_bdm, self.ephemeral_device = [(dev, 20 * GI * 1000)], dev


AWS_TO_KUBE_ARCH = {"x86_64": "amd64", "arm64": "arm64"}
_SCHEME = re.compile(r"(^[a-z]+)(\-[0-9]+tb)?([0-9]+).*\.")  # types.go:48


def lower_kabob(s):
    return s.lower().replace(" ", "-")


def get_architecture(info):  # types.go:184-191
    for a in info["ProcessorInfo"]["SupportedArchitectures"]:
        if a in AWS_TO_KUBE_ARCH:
            return AWS_TO_KUBE_ARCH[a]
    return "[" + " ".join(info["ProcessorInfo"]["SupportedArchitectures"]) + "]"


def cpu_milli(info):
    return info["VCpuInfo"]["DefaultVCpus"] * 1000


def memory_milli(info, vm_overhead_pct):  # types.go:216-226
    size_mib = info["MemoryInfo"]["SizeInMiB"]
    archs = info["ProcessorInfo"]["SupportedArchitectures"]
    if len(archs) > 0 and archs[0] == "arm64":
        size_mib -= 64
    mem_bytes = size_mib * MI
    overhead_mib = int(math.ceil(float(mem_bytes) * vm_overhead_pct / 1024 / 1024))
    return (mem_bytes - overhead_mib * MI) * 1000


def ephemeral_storage_milli(info, fam, ctx):  # types.go:229-264
    if ctx.get("instance_store_policy") == "RAID0":
        st = info.get("InstanceStorageInfo")
        if st is not None and st.get("TotalSizeInGB") is not None:
            return st["TotalSizeInGB"] * 10 ** 9 * 1000
    bdms = ctx.get("block_device_mappings") or []
    if bdms:
        for b in bdms:
            if b.get("root_volume") and b.get("volume_size") is not None:
                return b["volume_size"]
        if fam.custom:
            vs = bdms[-1].get("volume_size")
            return vs if vs is not None else 20 * GI * 1000
        for b in bdms:
            if b.get("device_name") == fam.ephemeral_device and b.get("volume_size") is not None:
                return b["volume_size"]
    for dev, size in fam.default_bdm:
        if dev == fam.ephemeral_device:
            return size
    return 20 * GI * 1000


def eni_limited_pods(info, reserved_enis):  # types.go:338-352
    ni = info["NetworkInfo"]
    cards = ni["NetworkCards"]
    card = cards[ni.get("DefaultNetworkCardIndex", 0)]
    usable = max(card["MaximumNetworkInterfaces"] - reserved_enis, 0)
    if usable == 0:
        return 0
    return usable * (ni["Ipv4AddressesPerInterface"] - 1) + 2


def pods_count(info, fam, ctx):  # types.go:433-448
    max_pods, ppc = ctx.get("max_pods"), ctx.get("pods_per_core")
    if max_pods is not None:
        count = int(max_pods)
    elif fam.flags["eni_pods"]:
        count = eni_limited_pods(info, ctx.get("reserved_enis", 0))
    else:
        count = 110
    if (ppc or 0) > 0 and fam.flags["pods_per_core"]:
        count = min(int(ppc) * info["VCpuInfo"]["DefaultVCpus"], count)
    return count


def neurons(info):  # types.go:302-316
    name = info["InstanceType"]
    if name == "trn1.2xlarge":
        return 1
    if name in ("trn1.32xlarge", "trn1n.32xlarge"):
        return 16
    acc = info.get("InferenceAcceleratorInfo")
    return sum(a["Count"] for a in acc["Accelerators"]) if acc else 0


def gpus_by(info, manufacturer):
    g = info.get("GpuInfo")
    return sum(x["Count"] for x in g["Gpus"] if x["Manufacturer"] == manufacturer) if g else 0


def compute_capacity(info, fam, ctx, limits):  # types.go:193-210
    name = info["InstanceType"]
    lim = limits.get(name)
    efa = 0
    ni = info.get("NetworkInfo")
    if ni is not None and ni.get("EfaInfo") is not None:
        efa = ni["EfaInfo"].get("MaximumEfaInterfaces") or 0
    return {
        RES_CPU: cpu_milli(info),
        RES_MEM: memory_milli(info, ctx.get("vm_memory_overhead_percent", 0.075)),
        RES_EPH: ephemeral_storage_milli(info, fam, ctx),
        RES_PODS: pods_count(info, fam, ctx) * 1000,
        RES_POD_ENI: (lim["branch_interface"] if lim and lim["trunking"] else 0) * 1000,
        RES_NVIDIA: gpus_by(info, "NVIDIA") * 1000,
        RES_AMD: gpus_by(info, "AMD") * 1000,
        RES_NEURON: neurons(info) * 1000,
        RES_GAUDI: gpus_by(info, "Habana") * 1000,
        RES_EFA: efa * 1000,
    }


def kube_reserved(cpu_m, pods, eni_pods, fam, overrides):  # types.go:369-402
    if fam.flags["eni_mem"]:
        pods = eni_pods
    res = {RES_MEM: (11 * pods + 255) * MI * 1000, RES_EPH: 1 * GI * 1000}
    cpu_overhead = 0
    for start, end, pct in ((0, 1000, 0.06), (1000, 2000, 0.01), (2000, 4000, 0.005), (4000, 1 << 31, 0.0025)):
        if cpu_m >= start:
            r = float(end - start)
            if cpu_m < end:
                r = float(cpu_m - start)
            cpu_overhead += int(r * pct)
            res[RES_CPU] = cpu_overhead
    for k, v in (overrides or {}).items():
        res[k] = parse_quantity(v)
    return res


def _eviction_signal(capacity_milli, value):  # types.go:456-478
    if value.endswith("%"):
        p = float(value.strip("%"))
        if p == 100:
            p = 0
        # capacity.AsApproximateFloat64() / 100 * p, then Ceil, printed and re-parsed (exact integers)
        return int(math.ceil(capacity_milli / 1000.0 / 100 * p)) * 1000
    return parse_quantity(value)


def eviction_threshold(mem_m, storage_m, fam, hard, soft):  # types.go:404-431
    overhead = {RES_MEM: 100 * MI * 1000, RES_EPH: int(math.ceil(float(storage_m // 1000) / 100 * 10)) * 1000}
    override = {}
    signals = []
    if hard is not None:
        signals.append(hard)
    if soft is not None and fam.flags["eviction_soft"]:
        signals.append(soft)
    for m in signals:
        temp = {}
        if "memory.available" in m:
            temp[RES_MEM] = _eviction_signal(mem_m, m["memory.available"])
        if "nodefs.available" in m:
            temp[RES_EPH] = _eviction_signal(storage_m, m["nodefs.available"])
        for k, v in temp.items():  # resources.MaxResources
            override[k] = max(override.get(k, v), v)
    overhead.update(override)
    return overhead


def create_offerings(info, all_zones, type_zones, ctx):  # instancetype.go:264-315
    """Canonical order: sorted zones, then sorted capacity types (Go iterates maps)."""
    name = info["InstanceType"]
    subnets = {s["zone"]: s.get("zone_id") for s in ctx["subnets"]}
    unavailable = ctx.get("unavailable") or set()
    od, spot = ctx["od_prices"], ctx.get("spot_prices") or {}
    out = []
    for zone in sorted(all_zones):
        for ct in sorted(set(info.get("SupportedUsageClasses") or [])):
            is_unavail = (ct, name, zone) in unavailable
            if ct == "spot":
                price = spot.get((name, zone))
                ok = price is not None
            elif ct == "on-demand":
                price = od.get(name)
                ok = price is not None
            else:
                continue  # capacity-block and unknown classes are skipped
            has_subnet = zone in subnets
            available = (not is_unavail) and ok and (zone in type_zones) and has_subnet
            zid = subnets.get(zone) or None
            out.append((zone, ct, zid, price if ok else 0.0, available))
    return out


def compute_requirements(info, offerings, fam, ctx, bandwidth):  # types.go:74-172
    name = info["InstanceType"]
    avail = [o for o in offerings if o[4]]
    dne = "DoesNotExist"
    arch = get_architecture(info)
    if fam.windows:
        os_vals = ["windows"] if arch == "amd64" else []
    else:
        os_vals = ["linux"]
    r = {
        L_INSTANCE_TYPE: ("In", [name]),
        L_ARCH: ("In", [arch]),
        L_OS: ("In", os_vals),
        L_ZONE: ("In", sorted({o[0] for o in avail})),
        L_REGION: ("In", [ctx.get("region", "us-east-1")]),
        L_WINDOWS_BUILD: (dne, []),
        L_CAPACITY_TYPE: ("In", sorted({o[1] for o in avail})),
        G + "/instance-cpu": ("In", [str(info["VCpuInfo"]["DefaultVCpus"])]),
        G + "/instance-cpu-manufacturer": (dne, []),
        G + "/instance-memory": ("In", [str(info["MemoryInfo"]["SizeInMiB"])]),
        G + "/instance-ebs-bandwidth": (dne, []),
        G + "/instance-network-bandwidth": (dne, []),
        G + "/instance-category": (dne, []),
        G + "/instance-family": (dne, []),
        G + "/instance-generation": (dne, []),
        G + "/instance-local-nvme": (dne, []),
        G + "/instance-size": (dne, []),
        G + "/instance-gpu-name": (dne, []),
        G + "/instance-gpu-manufacturer": (dne, []),
        G + "/instance-gpu-count": (dne, []),
        G + "/instance-gpu-memory": (dne, []),
        G + "/instance-accelerator-name": (dne, []),
        G + "/instance-accelerator-manufacturer": (dne, []),
        G + "/instance-accelerator-count": (dne, []),
        G + "/instance-hypervisor": ("In", [info.get("Hypervisor") or ""]),
        G + "/instance-encryption-in-transit-supported": ("In", ["true" if (info.get("NetworkInfo") or {}).get("EncryptionInTransitSupported") else "false"]),
    }

    def insert(key, value):  # Requirement.Insert on a DoesNotExist requirement turns it into In [value]
        op, vals = r[key]
        r[key] = ("In", sorted(set(vals) | {value}))

    zone_ids = sorted({o[2] for o in avail if o[2]})
    if zone_ids:
        r[L_ZONE_ID] = ("In", zone_ids)
    m = _SCHEME.search(name)
    if m:
        insert(G + "/instance-category", m.group(1))
        insert(G + "/instance-generation", m.group(3))
    parts = name.split(".")
    if len(parts) == 2:
        insert(G + "/instance-family", parts[0])
        insert(G + "/instance-size", parts[1])
    st = info.get("InstanceStorageInfo")
    if st is not None and st.get("NvmeSupport") != "unsupported":
        insert(G + "/instance-local-nvme", str(st.get("TotalSizeInGB") or 0))
    if name in bandwidth:
        insert(G + "/instance-network-bandwidth", str(bandwidth[name]))
    g = info.get("GpuInfo")
    if g is not None and len(g["Gpus"]) == 1:
        gpu = g["Gpus"][0]
        insert(G + "/instance-gpu-name", lower_kabob(gpu["Name"]))
        insert(G + "/instance-gpu-manufacturer", lower_kabob(gpu["Manufacturer"]))
        insert(G + "/instance-gpu-count", str(gpu["Count"]))
        insert(G + "/instance-gpu-memory", str(gpu["MemoryInfo"]["SizeInMiB"]))
    acc = info.get("InferenceAcceleratorInfo")
    if acc is not None and len(acc["Accelerators"]) == 1:
        a = acc["Accelerators"][0]
        insert(G + "/instance-accelerator-name", lower_kabob(a["Name"]))
        insert(G + "/instance-accelerator-manufacturer", lower_kabob(a["Manufacturer"]))
        insert(G + "/instance-accelerator-count", str(a["Count"]))
    if fam.windows:
        insert(L_WINDOWS_BUILD, fam.build)
    if name.startswith("trn1"):
        insert(G + "/instance-accelerator-name", lower_kabob("Inferentia"))
        insert(G + "/instance-accelerator-manufacturer", lower_kabob("AWS"))
        insert(G + "/instance-accelerator-count", str(neurons(info)))
    pi = info.get("ProcessorInfo")
    if pi is not None and pi.get("Manufacturer") is not None:
        insert(G + "/instance-cpu-manufacturer", lower_kabob(pi["Manufacturer"]))
    ebs = info.get("EbsInfo")
    if ebs is not None and ebs.get("EbsOptimizedSupport") == "default":
        insert(G + "/instance-ebs-bandwidth", str(ebs["EbsOptimizedInfo"]["MaximumBandwidthInMbps"]))
    return [(k, op, vals) for k, (op, vals) in r.items()]


def new_instance_type(info, ctx, limits, bandwidth, all_zones, type_zones):  # types.go:51-71
    fam = AMIFamily(ctx.get("ami_family", "AL2023"))
    offerings = create_offerings(info, all_zones, type_zones, ctx)
    reqs = compute_requirements(info, offerings, fam, ctx, bandwidth)
    cap = compute_capacity(info, fam, ctx, limits)
    kr = kube_reserved(cpu_milli(info), pods_count(info, fam, ctx), eni_limited_pods(info, ctx.get("reserved_enis", 0)),
                       fam, ctx.get("kube_reserved"))
    sr = {k: parse_quantity(v) for k, v in (ctx.get("system_reserved") or {}).items()}
    ev = eviction_threshold(cap[RES_MEM], cap[RES_EPH], fam, ctx.get("eviction_hard"), ctx.get("eviction_soft"))
    overhead = {}
    for part in (kr, sr, ev):
        for k, v in part.items():
            overhead[k] = overhead.get(k, 0) + v
    # types.go:67-69: windows-compatible types also advertise PrivateIPv4Address
    os_req = dict((k, (op, v)) for k, op, v in reqs)[L_OS]
    if "windows" in os_req[1]:
        lim = limits.get(info["InstanceType"])
        cap[RES_PRIVATE_IPV4] = ((lim["ipv4_per_interface"] - 1) if lim else 0) * 1000
    return {"name": info["InstanceType"], "requirements": reqs, "capacity": cap, "overhead": overhead,
            "offerings": offerings, "overhead_parts": {"kube": kr, "system": sr, "eviction": ev}}


def list_instance_types(infos, ctx, limits, bandwidth, offering_zones):
    """DefaultProvider.List miss path (instancetype.go:143-172)."""
    all_zones = set()
    for zs in offering_zones.values():
        all_zones.update(zs)
    return [new_instance_type(i, ctx, limits, bandwidth, all_zones, set(offering_zones.get(i["InstanceType"], ()))) for i in infos]


def allocatable(t):
    return {k: v - t["overhead"].get(k, 0) for k, v in t["capacity"].items()}


# --------------------------------------------------------------------------------------------
def info_from_doc_row(row, limits):
    """Rebuild the EC2 facts the doc generator saw for one of the 776 rows, from its labels plus
    the VPC limits table.  Returns None when a needed input is not recoverable."""
    lab = row["labels"]
    name = row["name"]
    lim = limits.get(name)
    if lim is None:
        return None
    arch = lab["kubernetes.io/arch"]
    info = {
        "InstanceType": name,
        "ProcessorInfo": {"SupportedArchitectures": ["arm64" if arch == "arm64" else "x86_64"]},
        "VCpuInfo": {"DefaultVCpus": int(lab[G + "/instance-cpu"])},
        "MemoryInfo": {"SizeInMiB": int(lab[G + "/instance-memory"])},
        "Hypervisor": lab.get(G + "/instance-hypervisor", ""),
        "SupportedUsageClasses": ["on-demand", "spot"],
        "NetworkInfo": {
            "DefaultNetworkCardIndex": lim["default_card"],
            "NetworkCards": [{"MaximumNetworkInterfaces": c[0], "NetworkCardIndex": c[1]} for c in lim["cards"]],
            "Ipv4AddressesPerInterface": lim["ipv4_per_interface"],
            "EncryptionInTransitSupported": lab.get(G + "/instance-encryption-in-transit-supported") == "true",
        },
    }
    if G + "/instance-cpu-manufacturer" in lab:
        info["ProcessorInfo"]["Manufacturer"] = lab[G + "/instance-cpu-manufacturer"]
    if G + "/instance-gpu-count" in lab:
        man = {"nvidia": "NVIDIA", "amd": "AMD", "habana": "Habana"}.get(lab[G + "/instance-gpu-manufacturer"], lab[G + "/instance-gpu-manufacturer"])
        info["GpuInfo"] = {"Gpus": [{"Name": lab[G + "/instance-gpu-name"], "Manufacturer": man,
                                     "Count": int(lab[G + "/instance-gpu-count"]),
                                     "MemoryInfo": {"SizeInMiB": int(lab[G + "/instance-gpu-memory"])}}]}
    if G + "/instance-accelerator-count" in lab and not name.startswith("trn1"):
        info["InferenceAcceleratorInfo"] = {"Accelerators": [{"Name": lab[G + "/instance-accelerator-name"],
                                                               "Manufacturer": lab[G + "/instance-accelerator-manufacturer"],
                                                               "Count": int(lab[G + "/instance-accelerator-count"])}]}
    if G + "/instance-local-nvme" in lab:
        info["InstanceStorageInfo"] = {"NvmeSupport": "required", "TotalSizeInGB": int(lab[G + "/instance-local-nvme"])}
    if G + "/instance-ebs-bandwidth" in lab:
        info["EbsInfo"] = {"EbsOptimizedSupport": "default", "EbsOptimizedInfo": {"MaximumBandwidthInMbps": int(lab[G + "/instance-ebs-bandwidth"])}}
    efa = row["resources"].get(RES_EFA)
    if efa is not None:  # EFA count is an input we cannot rebuild from labels; taken from the row itself
        info["NetworkInfo"]["EfaInfo"] = {"MaximumEfaInterfaces": int(efa)}
    return info
