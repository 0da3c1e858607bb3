// BenchmarkSolveC2: upstream's Scheduler.Solve on the C2 batch of bench.py (10 000 mixed pods x 776 instance types x 3 zones x
// spot + on-demand, one NodePool).  NOT COMPILED HERE (no Go toolchain, no module): see README.md.
package goref

import (
	"context"
	"encoding/json"
	"fmt"
	"os"
	"runtime"
	"testing"
	"time"

	v1 "k8s.io/api/core/v1"
	"k8s.io/apimachinery/pkg/api/resource"
	metav1 "k8s.io/apimachinery/pkg/apis/meta/v1"
	"k8s.io/apimachinery/pkg/types"
	"k8s.io/apimachinery/pkg/util/sets"
	"k8s.io/client-go/tools/record"
	"k8s.io/utils/clock"
	"sigs.k8s.io/controller-runtime/pkg/client/fake"

	corev1beta1 "sigs.k8s.io/karpenter/pkg/apis/v1beta1"
	"sigs.k8s.io/karpenter/pkg/cloudprovider"
	fakecp "sigs.k8s.io/karpenter/pkg/cloudprovider/fake"
	"sigs.k8s.io/karpenter/pkg/controllers/provisioning/scheduling"
	"sigs.k8s.io/karpenter/pkg/controllers/state"
	"sigs.k8s.io/karpenter/pkg/events"
	pscheduling "sigs.k8s.io/karpenter/pkg/scheduling"
)

// fixture written by tools/export_go_fixture.py
type fxReq struct {
	Key    string   `json:"key"`
	Op     string   `json:"op"`
	Values []string `json:"values"`
}
type fxOffering struct {
	Zone         string  `json:"zone"`
	CapacityType string  `json:"capacity_type"`
	Price        float64 `json:"price"`
	Available    bool    `json:"available"`
}
type fxType struct {
	Name         string            `json:"name"`
	Requirements []fxReq           `json:"requirements"`
	Offerings    []fxOffering      `json:"offerings"`
	Capacity     map[string]string `json:"capacity"`
	KubeReserved map[string]string `json:"kube_reserved"`
	SysReserved  map[string]string `json:"system_reserved"`
	Eviction     map[string]string `json:"eviction_threshold"`
}
type fxPod struct {
	UID          string            `json:"uid"`
	Requests     map[string]string `json:"requests"`
	NodeSelector map[string]string `json:"node_selector"`
	Required     [][]fxReq         `json:"required_terms"`
	CreationTS   int64             `json:"creation_ts"`
}
type fixture struct {
	Types []fxType `json:"types"`
	Pods  []fxPod  `json:"pods"`
}

func quantities(m map[string]string) v1.ResourceList {
	out := v1.ResourceList{}
	for k, q := range m {
		out[v1.ResourceName(k)] = resource.MustParse(q)
	}
	return out
}

func nodeSelectorReqs(rs []fxReq) []v1.NodeSelectorRequirement {
	out := make([]v1.NodeSelectorRequirement, 0, len(rs))
	for _, r := range rs {
		out = append(out, v1.NodeSelectorRequirement{Key: r.Key, Operator: v1.NodeSelectorOperator(r.Op), Values: r.Values})
	}
	return out
}

func instanceTypes(fx *fixture) []*cloudprovider.InstanceType {
	out := make([]*cloudprovider.InstanceType, 0, len(fx.Types))
	for _, t := range fx.Types {
		reqs := pscheduling.NewRequirements()
		for _, r := range t.Requirements {
			reqs.Add(pscheduling.NewRequirement(r.Key, v1.NodeSelectorOperator(r.Op), r.Values...))
		}
		var offs cloudprovider.Offerings
		for _, o := range t.Offerings {
			offs = append(offs, cloudprovider.Offering{Zone: o.Zone, CapacityType: o.CapacityType, Price: o.Price, Available: o.Available})
		}
		out = append(out, &cloudprovider.InstanceType{
			Name: t.Name, Requirements: reqs, Offerings: offs, Capacity: quantities(t.Capacity),
			Overhead: &cloudprovider.InstanceTypeOverhead{
				KubeReserved: quantities(t.KubeReserved), SystemReserved: quantities(t.SysReserved), EvictionThreshold: quantities(t.Eviction),
			},
		})
	}
	return out
}

func pods(fx *fixture) []*v1.Pod {
	out := make([]*v1.Pod, 0, len(fx.Pods))
	for _, p := range fx.Pods {
		pod := &v1.Pod{
			ObjectMeta: metav1.ObjectMeta{Name: p.UID, Namespace: "default", UID: types.UID(p.UID), Labels: map[string]string{"app": "mix"},
				CreationTimestamp: metav1.NewTime(time.Unix(p.CreationTS, 0))},
			Spec: v1.PodSpec{NodeSelector: p.NodeSelector, Containers: []v1.Container{{Name: "c", Image: "pause",
				Resources: v1.ResourceRequirements{Requests: quantities(p.Requests)}}}},
		}
		if len(p.Required) > 0 {
			var terms []v1.NodeSelectorTerm
			for _, t := range p.Required {
				terms = append(terms, v1.NodeSelectorTerm{MatchExpressions: nodeSelectorReqs(t)})
			}
			pod.Spec.Affinity = &v1.Affinity{NodeAffinity: &v1.NodeAffinity{
				RequiredDuringSchedulingIgnoredDuringExecution: &v1.NodeSelector{NodeSelectorTerms: terms}}}
		}
		out = append(out, pod)
	}
	return out
}

func BenchmarkSolveC2(b *testing.B) {
	path := os.Getenv("KG_GO_FIXTURE")
	if path == "" {
		path = "c2.json"
	}
	raw, err := os.ReadFile(path)
	if err != nil {
		b.Skipf("fixture %s: %v (python tools/export_go_fixture.py)", path, err)
	}
	var fx fixture
	if err := json.Unmarshal(raw, &fx); err != nil {
		b.Fatal(err)
	}
	ctx := context.Background()
	its := instanceTypes(&fx)
	nodePool := &corev1beta1.NodePool{ObjectMeta: metav1.ObjectMeta{Name: "default"}}
	nodePool.Spec.Template.Spec.Requirements = []corev1beta1.NodeSelectorRequirementWithMinValues{{
		NodeSelectorRequirement: v1.NodeSelectorRequirement{Key: corev1beta1.CapacityTypeLabelKey, Operator: v1.NodeSelectorOpIn,
			Values: []string{corev1beta1.CapacityTypeOnDemand, corev1beta1.CapacityTypeSpot}}}}
	cp := fakecp.NewCloudProvider()
	cp.InstanceTypes = its
	kube := fake.NewClientBuilder().Build()
	b.ReportAllocs()
	var elapsed time.Duration
	for i := 0; i < b.N; i++ {
		batch := pods(&fx) // (Solve mutates the pods it relaxes)
		cluster := state.NewCluster(clock.RealClock{}, kube, cp)
		topology, err := scheduling.NewTopology(ctx, kube, cluster, map[string]sets.Set[string]{}, batch)
		if err != nil {
			b.Fatal(err)
		}
		s := scheduling.NewScheduler(kube, []*scheduling.NodeClaimTemplate{scheduling.NewNodeClaimTemplate(nodePool)}, []corev1beta1.NodePool{*nodePool},
			cluster, nil, topology, map[string][]*cloudprovider.InstanceType{nodePool.Name: its}, nil, events.NewRecorder(&record.FakeRecorder{}))
		t0 := time.Now()
		res := s.Solve(ctx, batch)
		elapsed += time.Since(t0)
		if i == 0 {
			fmt.Printf("NodeClaims %d, pod errors %d\n", len(res.NewNodeClaims), len(res.PodErrors))
		}
	}
	fmt.Printf("C2: %d pods, %.0f pods/s (GOMAXPROCS %d, NumCPU %d, %d Solves)\n", len(fx.Pods),
		float64(len(fx.Pods)*b.N)/elapsed.Seconds(), runtime.GOMAXPROCS(0), runtime.NumCPU(), b.N)
}
