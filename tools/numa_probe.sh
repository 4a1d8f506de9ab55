python - <<'PY'
import torch, os, glob
p = torch.cuda.get_device_properties(0)
print("pci", getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None), getattr(p, "pci_domain_id", None))
for d in glob.glob("/sys/class/drm/card*/device/numa_node"):
    print(d, open(d).read().strip(), open(os.path.dirname(d) + "/uevent").read().split("PCI_SLOT_NAME=")[1].split()[0])
for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    print(n, open(n).read().strip())
print("affinity", len(os.sched_getaffinity(0)))
PY
lscpu | grep -i numa
