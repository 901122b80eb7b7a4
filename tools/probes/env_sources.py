"""Which telemetry source reports live clock / power for the visible device under load (amdsmi metrics, sysfs hwmon by PCI address, rocm-smi)?"""
import os, sys, time, threading, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
x = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
stop = False
def load():
    while not stop:
        for _ in range(50): (x @ x)
        torch.cuda.synchronize()
th = threading.Thread(target=load); th.start()
time.sleep(1.0)
pr = torch.cuda.get_device_properties(0)
print("pci", getattr(pr, "pci_domain_id", None), pr.pci_bus_id, pr.pci_device_id)
try:
    import amdsmi
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    print("amdsmi handles", len(hs))
    for h in hs[:2]:
        try: print("bdf", amdsmi.amdsmi_get_gpu_device_bdf(h))
        except Exception as e: print("bdf err", e)
        try:
            m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            print({k: m[k] for k in m if any(s in k for s in ("gfxclk", "socket_power", "temperature_hotspot", "average_gfx_activity", "current_uclk"))})
        except Exception as e: print("metrics err", e)
        try: print("power", amdsmi.amdsmi_get_power_info(h))
        except Exception as e: print("power err", e)
        try: print("clk", amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX))
        except Exception as e: print("clk err", e)
except Exception as e:
    print("amdsmi unavailable", e)
bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
node = "/sys/bus/pci/devices/" + bdf
print(node, os.path.isdir(node))
if os.path.isdir(node):
    import glob
    for f in glob.glob(node + "/hwmon/hwmon*/*_input") + glob.glob(node + "/hwmon/hwmon*/power1_average") + [node + "/pp_dpm_sclk", node + "/gpu_busy_percent"]:
        try: print(f.replace(node, ""), open(f).read().strip().replace("\n", " | "))
        except OSError as e: print(f.replace(node, ""), "ERR", e)
print(subprocess.run("rocm-smi --showclocks --showpower --showuse --json", shell=True, capture_output=True, text=True).stdout[:800])
stop = True; th.join()
