// catalog_ec2.cpp -- kg_catalog_build_ec2: the List() miss path done natively.
//
// From raw EC2 facts to cloudprovider.InstanceType rows, following the reference line by line:
//   pkg/providers/instancetype/instancetype.go:264-315  createOfferings
//   pkg/providers/instancetype/types.go:51-71           NewInstanceType
//   types.go:74-172    computeRequirements       types.go:193-336   computeCapacity and helpers
//   types.go:338-352   ENILimitedPods            types.go:363-431   system/kube reserved, evictionThreshold
//   types.go:433-478   pods, computeEvictionSignal
//   pkg/providers/amifamily/resolver.go:93-111, bottlerocket.go:137-143, windows.go:86-92  feature flags, default volumes
// The rows then go through the same table build as kg_catalog_upload (catalog.cpp).
#include <cmath>
#include <set>

#include "ctx.hpp"

namespace kg {
namespace {

constexpr int64_t MI = 1024 * 1024, GI = 1024 * MI;

struct Family {
  bool eni_mem, pods_per_core, eviction_soft, eni_pods, windows, custom;
  int64_t default_ephemeral_milli;
  const char* build;
};

Family family_of(uint8_t f) {
  switch (f) {
    case KG_AMI_BOTTLEROCKET: return {false, false, false, true, false, false, 20 * GI * 1000, nullptr};
    case KG_AMI_WINDOWS2019: return {false, true, true, false, true, false, 50 * GI * 1000, "10.0.17763"};
    case KG_AMI_WINDOWS2022: return {false, true, true, false, true, false, 50 * GI * 1000, "10.0.20348"};
    case KG_AMI_CUSTOM: return {true, true, true, true, false, true, 20 * GI * 1000, nullptr};
    default: return {true, true, true, true, false, false, 20 * GI * 1000, nullptr};
  }
}

std::string lower_kabob(std::string_view s) {
  std::string o(s);
  for (auto& c : o) { if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a'); if (c == ' ') c = '-'; }
  return o;
}

// instanceTypeScheme = (^[a-z]+)(\-[0-9]+tb)?([0-9]+).*\.   (types.go:48)
bool scheme(const std::string& name, std::string* category, std::string* generation) {
  size_t i = 0;
  while (i < name.size() && name[i] >= 'a' && name[i] <= 'z') ++i;
  if (i == 0) return false;
  auto digits_then_dot = [&](size_t from, std::string* gen) {
    size_t j = from;
    while (j < name.size() && name[j] >= '0' && name[j] <= '9') ++j;
    if (j == from) return false;
    if (name.find('.', j) == std::string::npos) return false;
    *gen = name.substr(from, j - from);
    return true;
  };
  if (i < name.size() && name[i] == '-') {
    size_t j = i + 1;
    while (j < name.size() && name[j] >= '0' && name[j] <= '9') ++j;
    if (j > i + 1 && name.compare(j, 2, "tb") == 0 && digits_then_dot(j + 2, generation)) { *category = name.substr(0, i); return true; }
  }
  if (digits_then_dot(i, generation)) { *category = name.substr(0, i); return true; }
  return false;
}

// resource.MustParse -> int64 milli, in exact integer arithmetic: sign, digits, optional fraction, then a binary (Ki..Ei),
// decimal (n u m k M G T P E) or exponent (e3 / E-3) suffix.  A value finer than one milli-unit is not representable in the
// library's quantities: KG_EUNSUPPORTED (reject, do not round), as is one beyond int64.
int64_t parse_quantity_milli(std::string_view s) {
  using i128 = __int128;
  size_t i = 0;
  bool neg = false;
  if (i < s.size() && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; ++i; }
  i128 mant = 0;      // the digits, fraction included
  int frac = 0;       // how many of them follow the point
  bool any = false, point = false;
  for (; i < s.size(); ++i) {
    if (s[i] == '.' && !point) { point = true; continue; }
    if (s[i] < '0' || s[i] > '9') break;
    mant = mant * 10 + (s[i] - '0');
    if (point) ++frac;
    any = true;
    if (mant > (i128)1 << 100) throw Unsupported("quantity with too many digits");
  }
  kgdesc::check(any, "empty quantity");
  const std::string suf(s.substr(i));
  i128 num = 1000, den = 1;  // value = mant / 10^frac * num / den milli-units
  auto pow10 = [](int e) { i128 p = 1; for (int k = 0; k < e; ++k) p *= 10; return p; };
  if (suf.empty()) {}
  else if (suf == "n") den = pow10(9);
  else if (suf == "u") den = pow10(6);
  else if (suf == "m") num = 1;
  else if (suf == "k") num = 1000 * pow10(3);
  else if (suf == "M") num = 1000 * pow10(6);
  else if (suf == "G") num = 1000 * pow10(9);
  else if (suf == "T") num = 1000 * pow10(12);
  else if (suf == "P") num = 1000 * pow10(15);
  else if (suf == "E") num = 1000 * pow10(18);
  else if (suf == "Ki") num = (i128)1000 << 10;
  else if (suf == "Mi") num = (i128)1000 << 20;
  else if (suf == "Gi") num = (i128)1000 << 30;
  else if (suf == "Ti") num = (i128)1000 << 40;
  else if (suf == "Pi") num = (i128)1000 << 50;
  else if (suf == "Ei") num = (i128)1000 << 60;
  else if ((suf[0] == 'e' || suf[0] == 'E') && suf.size() > 1) {
    size_t k = 1;
    bool eneg = false;
    if (suf[k] == '+' || suf[k] == '-') { eneg = suf[k] == '-'; ++k; }
    int e = 0;
    kgdesc::check(k < suf.size(), "quantity exponent");
    for (; k < suf.size(); ++k) { kgdesc::check(suf[k] >= '0' && suf[k] <= '9' && e < 100, "quantity exponent"); e = e * 10 + (suf[k] - '0'); }
    if (e > 24) throw Unsupported("quantity exponent out of range");
    if (eneg) den = pow10(e); else num = 1000 * pow10(e);
  } else throw Unsupported("quantity suffix not handled by the native catalog build");
  if (frac > 30) throw Unsupported("quantity finer than one milli-unit");
  constexpr i128 I128_MAX = (i128)(~(unsigned __int128)0 >> 1);
  {
    // (every product is checked before it is formed: mant < 2^100, num <= 10^27, den <= 10^24 * 10^30 would overflow 128 bits)
    const i128 f = pow10(frac);
    if (den > I128_MAX / f) {  // the divisor alone exceeds anything mant * num can reach here (num is 1000 when den > 1)
      if (mant == 0) return 0;
      throw Unsupported("quantity finer than one milli-unit");
    }
    den *= f;
  }
  if (mant != 0 && num > I128_MAX / mant) throw Unsupported("quantity beyond int64 milli-units");
  const i128 scaled = mant * num;
  if (scaled % den != 0) throw Unsupported("quantity finer than one milli-unit");
  const i128 v = scaled / den;
  if (v > (i128)INT64_MAX) throw Unsupported("quantity beyond int64 milli-units");
  return neg ? -(int64_t)v : (int64_t)v;
}

// computeEvictionSignal (types.go:456-478)
int64_t eviction_signal(int64_t capacity_milli, std::string_view value) {
  if (!value.empty() && value.back() == '%') {
    double p = std::stod(std::string(value.substr(0, value.size() - 1)));
    if (p == 100) p = 0;
    return (int64_t)std::ceil((double)capacity_milli / 1000.0 / 100 * p) * 1000;
  }
  return parse_quantity_milli(value);
}

void add_rl(std::map<uint32_t, int64_t>& m, uint32_t r, int64_t v) { m[r] += v; }

}  // namespace

void build_ec2_catalog(kg_ctx* ctx, const kg_ec2_catalog_desc* d, kg_catalog* cat) {
  Vocab& V = ctx->V;
  const kg_pools& p = d->pools;
  auto S = [&](uint32_t id) { return std::string(kgdesc::str(p.strings, id)); };
  const Family fam = family_of(d->ami_family);
  const char* Gp = "karpenter.k8s.aws/";
  auto key = [&](const std::string& k) { return V.key(k); };
  const uint32_t kIT = ctx->kInstanceType, kArch = key("kubernetes.io/arch"), kOS = key("kubernetes.io/os"), kZone = ctx->kZone,
                 kRegion = key("topology.kubernetes.io/region"), kWinBuild = key("node.kubernetes.io/windows-build"), kCT = ctx->kCT, kZid = ctx->kZid;
  auto gk = [&](const char* suffix) { return key(std::string(Gp) + suffix); };
  const uint32_t kCpu = gk("instance-cpu"), kCpuMan = gk("instance-cpu-manufacturer"), kMem = gk("instance-memory"), kEbs = gk("instance-ebs-bandwidth"),
                 kNet = gk("instance-network-bandwidth"), kCat = gk("instance-category"), kFam = gk("instance-family"), kGen = gk("instance-generation"),
                 kNvme = gk("instance-local-nvme"), kSize = gk("instance-size"), kGpuName = gk("instance-gpu-name"),
                 kGpuMan = gk("instance-gpu-manufacturer"), kGpuCount = gk("instance-gpu-count"), kGpuMem = gk("instance-gpu-memory"),
                 kAccName = gk("instance-accelerator-name"), kAccMan = gk("instance-accelerator-manufacturer"), kAccCount = gk("instance-accelerator-count"),
                 kHyp = gk("instance-hypervisor"), kEnc = gk("instance-encryption-in-transit-supported");
  auto R = [&](const char* n) { return V.resources.id(n); };
  const uint32_t rCPU = R("cpu"), rMem = R("memory"), rEph = R("ephemeral-storage"), rPods = R("pods"), rPodENI = R("vpc.amazonaws.com/pod-eni"),
                 rNvidia = R("nvidia.com/gpu"), rAmd = R("amd.com/gpu"), rNeuron = R("aws.amazon.com/neuron"), rGaudi = R("habana.ai/gaudi"),
                 rEfa = R("vpc.amazonaws.com/efa"), rIPv4 = R("vpc.amazonaws.com/PrivateIPv4Address");

  // provider state
  std::map<std::string, std::string> subnet_zone_id;  // zone -> zone id ("" when unknown)
  for (uint32_t i = 0; i < d->n_subnets; ++i) subnet_zone_id[S(d->subnets[i].zone)] = d->subnets[i].zone_id == KG_NONE ? "" : S(d->subnets[i].zone_id);
  std::map<std::string, double> od;
  for (uint32_t i = 0; i < d->n_od_prices; ++i) od[S(d->od_prices[i].instance_type)] = d->od_prices[i].price;
  std::map<std::pair<std::string, std::string>, double> spot;
  for (uint32_t i = 0; i < d->n_spot_prices; ++i) spot[{S(d->spot_prices[i].instance_type), S(d->spot_prices[i].zone)}] = d->spot_prices[i].price;
  std::set<std::tuple<std::string, std::string, std::string>> ice;
  for (uint32_t i = 0; i < d->n_unavailable; ++i) ice.insert({S(d->unavailable[i].capacity_type), S(d->unavailable[i].instance_type), S(d->unavailable[i].zone)});
  std::set<std::string> all_zones;  // every zone some type is offered in (instancetype.go:143-150)
  for (uint32_t i = 0; i < d->n_types; ++i)
    for (uint32_t z : kgdesc::values(p, d->types[i].zones_offered)) all_zones.insert(S(z));
  const std::string region = d->region == KG_NONE ? "" : S(d->region);
  auto rl_of = [&](kg_range r) { return parse_rl(V, p, r); };
  auto kube_override = rl_of(d->kube_reserved);
  auto system_reserved = rl_of(d->system_reserved);
  std::map<std::string, std::string> ev_hard, ev_soft;
  for (auto& l : kgdesc::labels(p, d->eviction_hard)) ev_hard[S(l.key)] = S(l.value);
  for (auto& l : kgdesc::labels(p, d->eviction_soft)) ev_soft[S(l.key)] = S(l.value);

  for (uint32_t ti = 0; ti < d->n_types; ++ti) {
    const kg_ec2_type& e = d->types[ti];
    CatType ct;
    ct.name = S(e.name);
    // ---- createOfferings (canonical order: zones, then usage classes, both sorted)
    std::set<std::string> type_zones, classes;
    for (uint32_t z : kgdesc::values(p, e.zones_offered)) type_zones.insert(S(z));
    for (uint32_t c : kgdesc::values(p, e.usage_classes)) classes.insert(S(c));
    std::set<std::string> avail_zones, avail_cts, avail_zids;
    for (auto& zone : all_zones) {
      for (auto& cls : classes) {
        double price = 0;
        bool ok = false;
        if (cls == "spot") { auto f = spot.find({ct.name, zone}); if (f != spot.end()) { price = f->second; ok = true; } }
        else if (cls == "on-demand") { auto f = od.find(ct.name); if (f != od.end()) { price = f->second; ok = true; } }
        else continue;
        auto sn = subnet_zone_id.find(zone);
        const bool available = !ice.count({cls, ct.name, zone}) && ok && type_zones.count(zone) && sn != subnet_zone_id.end();
        CatOffering o;
        o.zone = V.val(kZone, zone);
        o.ct = V.val(kCT, cls);
        o.zid = (sn != subnet_zone_id.end() && !sn->second.empty()) ? (int)V.val(kZid, sn->second) : -1;
        o.price = ok ? price : 0.0;
        o.available = available;
        ct.offerings.push_back(o);
        if (available) { avail_zones.insert(zone); avail_cts.insert(cls); if (o.zid >= 0) avail_zids.insert(sn->second); }
      }
    }
    // ---- computeRequirements
    auto in = [&](uint32_t k, const std::vector<std::string>& vals) {
      std::vector<uint32_t> ids;
      for (auto& v : vals) ids.push_back(V.val(k, v));
      ct.reqs.add(V, hreq_new(V, k, KG_OP_IN, ids, -1));
    };
    const std::string arch = S(e.arch);
    in(kIT, {ct.name});
    in(kArch, {arch});
    if (fam.windows) in(kOS, arch == "amd64" ? std::vector<std::string>{"windows"} : std::vector<std::string>{});
    else in(kOS, {"linux"});
    in(kZone, {avail_zones.begin(), avail_zones.end()});
    in(kRegion, {region});
    in(kCT, {avail_cts.begin(), avail_cts.end()});
    in(kCpu, {std::to_string(e.vcpus)});
    in(kMem, {std::to_string(e.memory_mib)});
    in(kHyp, {e.hypervisor == KG_NONE ? std::string() : S(e.hypervisor)});
    in(kEnc, {e.encryption_in_transit ? "true" : "false"});
    if (!avail_zids.empty()) in(kZid, {avail_zids.begin(), avail_zids.end()});
    std::string category, generation;
    const bool sch = scheme(ct.name, &category, &generation);
    in(kCat, sch ? std::vector<std::string>{category} : std::vector<std::string>{});
    in(kGen, sch ? std::vector<std::string>{generation} : std::vector<std::string>{});
    const size_t dot = ct.name.find('.');
    const bool two = dot != std::string::npos && ct.name.find('.', dot + 1) == std::string::npos;
    in(kFam, two ? std::vector<std::string>{ct.name.substr(0, dot)} : std::vector<std::string>{});
    in(kSize, two ? std::vector<std::string>{ct.name.substr(dot + 1)} : std::vector<std::string>{});
    in(kNvme, e.local_nvme_gb >= 0 ? std::vector<std::string>{std::to_string(e.local_nvme_gb)} : std::vector<std::string>{});
    in(kNet, e.network_bandwidth_mbps >= 0 ? std::vector<std::string>{std::to_string(e.network_bandwidth_mbps)} : std::vector<std::string>{});
    in(kEbs, e.ebs_max_bandwidth_mbps >= 0 ? std::vector<std::string>{std::to_string(e.ebs_max_bandwidth_mbps)} : std::vector<std::string>{});
    in(kCpuMan, e.cpu_manufacturer != KG_NONE ? std::vector<std::string>{lower_kabob(S(e.cpu_manufacturer))} : std::vector<std::string>{});
    in(kWinBuild, fam.windows ? std::vector<std::string>{fam.build} : std::vector<std::string>{});
    auto gpus = kgdesc::span(d->devices, d->n_devices, e.gpus, "gpu range");
    auto accs = kgdesc::span(d->devices, d->n_devices, e.accelerators, "accelerator range");
    const bool one_gpu = gpus.size() == 1;
    in(kGpuName, one_gpu ? std::vector<std::string>{lower_kabob(S(gpus[0].name))} : std::vector<std::string>{});
    in(kGpuMan, one_gpu ? std::vector<std::string>{lower_kabob(S(gpus[0].manufacturer))} : std::vector<std::string>{});
    in(kGpuCount, one_gpu ? std::vector<std::string>{std::to_string(gpus[0].count)} : std::vector<std::string>{});
    in(kGpuMem, one_gpu ? std::vector<std::string>{std::to_string(gpus[0].memory_mib)} : std::vector<std::string>{});
    int64_t neurons = 0;
    if (ct.name == "trn1.2xlarge") neurons = 1;
    else if (ct.name == "trn1.32xlarge" || ct.name == "trn1n.32xlarge") neurons = 16;
    else for (auto& a : accs) neurons += a.count;
    const bool trn1 = ct.name.rfind("trn1", 0) == 0;
    if (trn1) {  // types.go:158-162
      in(kAccName, {"inferentia"});
      in(kAccMan, {"aws"});
      in(kAccCount, {std::to_string(neurons)});
    } else {
      const bool one_acc = accs.size() == 1;
      in(kAccName, one_acc ? std::vector<std::string>{lower_kabob(S(accs[0].name))} : std::vector<std::string>{});
      in(kAccMan, one_acc ? std::vector<std::string>{lower_kabob(S(accs[0].manufacturer))} : std::vector<std::string>{});
      in(kAccCount, one_acc ? std::vector<std::string>{std::to_string(accs[0].count)} : std::vector<std::string>{});
    }
    // ---- computeCapacity
    std::map<uint32_t, int64_t> cap, ovh;
    const int64_t cpu_m = e.vcpus * 1000;
    const int64_t mem_bytes = (e.memory_mib - (e.first_arch_is_arm64 ? 64 : 0)) * MI;
    const int64_t vm_overhead_mib = (int64_t)std::ceil((double)mem_bytes * d->vm_memory_overhead_percent / 1024 / 1024);
    const int64_t mem_m = (mem_bytes - vm_overhead_mib * MI) * 1000;
    int64_t eph_m;
    if (d->instance_store_policy_raid0 && e.instance_storage_total_gb >= 0) eph_m = e.instance_storage_total_gb * 1000000000LL * 1000;
    else if (d->has_block_device_mappings && d->root_volume_milli >= 0) eph_m = d->root_volume_milli;
    else if (d->has_block_device_mappings && fam.custom) eph_m = d->last_bdm_volume_milli >= 0 ? d->last_bdm_volume_milli : 20 * GI * 1000;
    else if (d->has_block_device_mappings && d->ephemeral_device_volume_milli >= 0) eph_m = d->ephemeral_device_volume_milli;
    else eph_m = fam.default_ephemeral_milli;
    const int64_t usable = std::max<int64_t>(e.default_card_max_interfaces - d->reserved_enis, 0);
    const int64_t eni_pods = usable == 0 ? 0 : usable * (e.ipv4_per_interface - 1) + 2;
    int64_t pods;
    if (d->max_pods >= 0) pods = d->max_pods;
    else if (fam.eni_pods) pods = eni_pods;
    else pods = 110;
    if (d->pods_per_core > 0 && fam.pods_per_core) pods = std::min<int64_t>((int64_t)d->pods_per_core * e.vcpus, pods);
    auto by_man = [&](const char* man) { int64_t n = 0; for (auto& g : gpus) if (S(g.manufacturer) == man) n += g.count; return n; };
    cap[rCPU] = cpu_m; cap[rMem] = mem_m; cap[rEph] = eph_m; cap[rPods] = pods * 1000;
    cap[rPodENI] = (e.has_limits && e.trunking_compatible ? e.branch_interfaces : 0) * 1000;
    cap[rNvidia] = by_man("NVIDIA") * 1000; cap[rAmd] = by_man("AMD") * 1000; cap[rNeuron] = neurons * 1000; cap[rGaudi] = by_man("Habana") * 1000;
    cap[rEfa] = e.efa_max_interfaces * 1000;
    if (fam.windows && arch == "amd64") cap[rIPv4] = (e.has_limits ? e.limits_ipv4_per_interface - 1 : 0) * 1000;  // types.go:67-69
    // ---- overhead: kubeReserved + systemReserved + evictionThreshold
    {
      const int64_t kr_pods = fam.eni_mem ? eni_pods : pods;
      std::map<uint32_t, int64_t> kr;
      kr[rMem] = (11 * kr_pods + 255) * MI * 1000;
      kr[rEph] = 1 * GI * 1000;
      struct Tier { int64_t start, end; double pct; };
      const Tier tiers[] = {{0, 1000, 0.06}, {1000, 2000, 0.01}, {2000, 4000, 0.005}, {4000, int64_t(1) << 31, 0.0025}};
      int64_t cpu_overhead = 0;
      for (auto& t : tiers)
        if (cpu_m >= t.start) {
          double r = (double)(t.end - t.start);
          if (cpu_m < t.end) r = (double)(cpu_m - t.start);
          cpu_overhead += (int64_t)(r * t.pct);
          kr[rCPU] = cpu_overhead;
        }
      for (auto& kv : kube_override) kr[kv.first] = kv.second;
      for (auto& kv : kr) add_rl(ovh, kv.first, kv.second);
      for (auto& kv : system_reserved) add_rl(ovh, kv.first, kv.second);
      std::map<uint32_t, int64_t> ev;
      ev[rMem] = 100 * MI * 1000;
      ev[rEph] = (int64_t)std::ceil((double)(eph_m / 1000) / 100 * 10) * 1000;
      std::map<uint32_t, int64_t> override_;
      auto apply = [&](const std::map<std::string, std::string>& m) {
        std::map<uint32_t, int64_t> temp;
        auto f = m.find("memory.available");
        if (f != m.end()) temp[rMem] = eviction_signal(mem_m, f->second);
        f = m.find("nodefs.available");
        if (f != m.end()) temp[rEph] = eviction_signal(eph_m, f->second);
        for (auto& kv : temp) { auto o = override_.find(kv.first); if (o == override_.end() || kv.second > o->second) override_[kv.first] = kv.second; }
      };
      if (d->has_eviction_hard) apply(ev_hard);
      if (d->has_eviction_soft && fam.eviction_soft) apply(ev_soft);
      for (auto& kv : override_) ev[kv.first] = kv.second;
      for (auto& kv : ev) add_rl(ovh, kv.first, kv.second);
    }
    ct.capacity.assign(cap.begin(), cap.end());
    ct.overhead.assign(ovh.begin(), ovh.end());
    cat->types.push_back(std::move(ct));
  }
}

}  // namespace kg

extern "C" int32_t kg_catalog_build_ec2(kg_ctx* ctx, const kg_ec2_catalog_desc* desc, uint64_t generation, kg_catalog** out) {
  if (!ctx || !desc || !out) return KG_EINVAL;
  kg::CtxLock lk(ctx);
  std::unique_ptr<kg_catalog> cat(new kg_catalog());
  try {
    cat->ctx = ctx;
    cat->generation = generation;
    kg::build_ec2_catalog(ctx, desc, cat.get());
    kg::build_catalog_tables(ctx, cat.get());
    *out = cat.release();
    return KG_OK;
  } catch (const kg::Unsupported& e) { ctx->last_error = e.what(); return KG_EUNSUPPORTED; }
  catch (const kgdesc::DescError& e) { ctx->last_error = e.what(); return KG_EINVAL; }
  catch (const std::bad_alloc&) { ctx->last_error = "out of host memory"; return KG_ENOMEM; }
  catch (const std::exception& e) { ctx->last_error = e.what(); return KG_EDEVICE; }
}
