"""Package power / shader clock sampler for MI355X (diagnostics only; nothing on the product path imports this).

Sources, first one that works: the amdsmi Python binding, the amdgpu hwmon sysfs files, `rocm-smi --json`.
    with PowerSampler(period=0.1) as ps:
        ... run the workload, synchronise ...
    ps.summary()  ->  {"source": ..., "n": samples, "power_w": mean, "power_max_w": ..., "sclk_mhz": mean, ...}
"""
import glob
import json
import subprocess
import threading
import time


class _AmdSmi:
    name = "amdsmi"

    def __init__(self, index=0):
        import amdsmi
        self.a = amdsmi
        amdsmi.amdsmi_init()
        self.h = amdsmi.amdsmi_get_processor_handles()[index]
        self.read()   # raises when the binding cannot talk to the driver

    def read(self):
        a = self.a
        out = {}
        try:
            m = a.amdsmi_get_gpu_metrics_info(self.h)
            for k_src, k_dst in (("current_socket_power", "power_w"), ("average_socket_power", "power_w"),
                                 ("current_gfxclk", "sclk_mhz"), ("average_gfxclk_frequency", "sclk_mhz"),
                                 ("temperature_hotspot", "temp_c"), ("average_gfx_activity", "busy_pct")):
                v = m.get(k_src)
                if isinstance(v, (int, float)) and k_dst not in out and 0 < v < 60000:
                    out[k_dst] = float(v)
            cg = m.get("current_gfxclks")
            if isinstance(cg, (list, tuple)):
                vals = [float(x) for x in cg if isinstance(x, (int, float)) and 0 < x < 60000]
                if vals:
                    out["sclk_mhz"] = sum(vals) / len(vals)      # mean over the XCDs
        except Exception:
            pass
        if "power_w" not in out:
            p = a.amdsmi_get_power_info(self.h)
            for k in ("current_socket_power", "average_socket_power", "socket_power"):
                v = p.get(k)
                if isinstance(v, (int, float)) and v > 0:
                    out["power_w"] = float(v)
                    break
        if "sclk_mhz" not in out:
            c = a.amdsmi_get_clock_info(self.h, a.AmdSmiClkType.GFX)
            v = c.get("clk", c.get("cur_clk"))
            if isinstance(v, (int, float)):
                out["sclk_mhz"] = float(v)
        if "power_w" not in out:
            raise RuntimeError("amdsmi returned no power figure")
        return out


class _Sysfs:
    name = "sysfs"

    def __init__(self, index=0):
        self.power = None
        for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average",
                    "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
            c = sorted(glob.glob(pat))
            if c:
                self.power = c[min(index, len(c) - 1)]
                break
        if self.power is None:
            raise RuntimeError("no amdgpu hwmon power file")
        d = self.power.rsplit("/", 1)[0]
        self.freq = d + "/freq1_input"
        self.read()

    def read(self):
        out = {"power_w": int(open(self.power).read()) / 1e6}
        try:
            out["sclk_mhz"] = int(open(self.freq).read()) / 1e6
        except Exception:
            pass
        return out


class _RocmSmi:
    name = "rocm-smi"

    def __init__(self, index=0):
        self.read()

    def read(self):
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout)
        card = d[sorted(d)[0]]
        out = {}
        for k, v in card.items():
            kl = k.lower()
            if "power" in kl and "w" in kl and "power_w" not in out:
                try:
                    out["power_w"] = float(v)
                except ValueError:
                    pass
            if kl.startswith("sclk clock speed"):
                try:
                    out["sclk_mhz"] = float(str(v).strip("()Mhz "))
                except ValueError:
                    pass
        if "power_w" not in out:
            raise RuntimeError("rocm-smi gave no power figure")
        return out


def open_source(index=0):
    errs = []
    for cls in (_AmdSmi, _Sysfs, _RocmSmi):
        try:
            return cls(index)
        except Exception as e:   # noqa: BLE001
            errs.append(f"{cls.name}: {type(e).__name__}: {e}")
    raise RuntimeError("no power source available: " + "; ".join(errs))


class PowerSampler:
    def __init__(self, period=0.1, index=0, source=None):
        self.src = source or open_source(index)
        self.period = period
        self.samples = []
        self._stop = threading.Event()
        self._th = None

    def __enter__(self):
        self.samples = []
        self._stop.clear()
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()
        return self

    def _run(self):
        while not self._stop.is_set():
            try:
                s = self.src.read()
                s["t"] = time.perf_counter()
                self.samples.append(s)
            except Exception:   # noqa: BLE001
                pass
            self._stop.wait(self.period)

    def __exit__(self, *exc):
        self._stop.set()
        self._th.join()

    def summary(self, skip_s=0.5):
        """Mean / max over the samples, dropping the first `skip_s` seconds (ramp-up)."""
        ss = self.samples
        if ss:
            t0 = ss[0]["t"]
            kept = [s for s in ss if s["t"] - t0 >= skip_s] or ss
        else:
            kept = []
        out = {"source": self.src.name, "n": len(kept)}
        for k in ("power_w", "sclk_mhz", "temp_c", "busy_pct"):
            v = [s[k] for s in kept if k in s]
            if v:
                out[k] = round(sum(v) / len(v), 1)
                out[k.replace("_w", "_max_w") if k == "power_w" else k + "_min"] = round(max(v) if k == "power_w" else min(v), 1)
        return out


if __name__ == "__main__":
    src = open_source()
    print(src.name, src.read())
