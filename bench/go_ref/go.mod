module karpgpu/bench/go_ref

go 1.22

// the versions the reference pins (/root/reference/go.mod)
require (
	k8s.io/api v0.30.2
	k8s.io/apimachinery v0.30.2
	k8s.io/client-go v0.30.2
	k8s.io/utils v0.0.0-20240102154912-e7106e64919e
	sigs.k8s.io/controller-runtime v0.18.4
	sigs.k8s.io/karpenter v0.37.1-0.20240715214311-a122e25df3de
)
