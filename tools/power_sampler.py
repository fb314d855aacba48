"""The card's power sensor, power cap and shader-clock level while launches run (sysfs hwmon / pp_dpm_sclk of the PCI address this
process computes on; rocm-smi as the fallback).  Used by tools/chain_bench.py --power and tools/probe_energy.py."""
import json
import os


class PowerSampler:
    """Average socket power (W), the cap and the current shader clock of card 0, polled from sysfs while launches run."""

    def __init__(self):
        import ctypes
        import glob
        sensors = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") +
                         glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
        # the card this process computes on, by its PCI address (a host shows every card and partition of the node in sysfs)
        self.bdf = None
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
                self.bdf = buf.value.decode().lower()
        except OSError:
            pass
        mine = [f for f in sensors if self.bdf and os.path.realpath(f.split("/hwmon/")[0]).lower().endswith(self.bdf)]
        self.others = [f for f in sensors if f not in mine]
        self.power = mine or sensors
        self.cap = [f.rsplit("/", 1)[0] + "/power1_cap" for f in self.power]
        self.sclk = [f.split("/hwmon/")[0] + "/pp_dpm_sclk" for f in self.power]
        self.samples, self.clocks, self.stop = [], [], False
        self.other_samples = {f: [] for f in self.others}

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return f.read()
        except OSError:
            return None

    def _smi(self):
        import subprocess
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, text=True, timeout=20).stdout
            card = json.loads(out).get("card0", {})
            for k, v in card.items():
                if "power" in k.lower() and "(w)" in k.lower():
                    self.samples.append(float(v))
                if k.lower().startswith("sclk"):
                    self.clocks.append(float(str(v).strip("()Mhz ")))
        except Exception as e:                                             # noqa: BLE001  (a diagnostic tool: report, go on)
            self.error = repr(e)

    def run(self):
        import time
        while not self.stop:
            got = False
            for f, c in zip(self.power[:1], self.sclk[:1]):
                v = self._read(f)
                if v and v.strip().isdigit() and int(v) > 0:
                    self.samples.append(int(v) / 1e6)
                    got = True
                clk = self._read(c) or ""
                for line in clk.splitlines():
                    if line.rstrip().endswith("*"):
                        self.clocks.append(float(line.split(":")[1].replace("Mhz", "").replace("*", "").strip()))
            if not got:
                self._smi()
            for f in self.others:                   # (every other sensor of the host too: which card the load shows on)
                v = self._read(f)
                if v and v.strip().isdigit():
                    self.other_samples[f].append(int(v) / 1e6)
            time.sleep(0.05)

    def summary(self):
        cap = self._read(self.cap[0]) if self.cap else None
        s, c = sorted(self.samples), self.clocks
        others = {f.split("/")[4]: round(sum(v) / len(v), 1) for f, v in self.other_samples.items() if v}
        return {"pci": self.bdf, "sensor": (self.power[0] if self.power else "rocm-smi"), "samples": len(s),
                "other_cards_W_mean": others,
                "power_W_mean": round(sum(s) / len(s), 1) if s else None, "power_W_median": s[len(s) // 2] if s else None,
                "power_W_max": s[-1] if s else None, "power_cap_W": int(cap) / 1e6 if cap and cap.strip().isdigit() else None,
                "sclk_MHz_mean": round(sum(c) / len(c), 1) if c else None, "error": getattr(self, "error", None)}
