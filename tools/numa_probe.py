import torch, os, glob
p = torch.cuda.get_device_properties(0)
bus = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
print(bus)
d = f"/sys/bus/pci/devices/{bus}"
print(open(d + "/numa_node").read().strip(), open(d + "/local_cpulist").read().strip())
