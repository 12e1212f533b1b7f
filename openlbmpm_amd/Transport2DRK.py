"""Driver counterpart of the reference's RKCG2D/Transport2DRK.py (own code): tracers carried by the
colour-gradient two-phase flow, class Transport2DRK(pathIniFile).runTransport2DMPMCRKNew().

The reference file does not parse and its `transportsetup.ini` is not
shipped; key names are the ones its constructor reads (Transport2DRK.py:31-311,
openlbmpm_amd/config.read_transport).  Kept: the flow set-up of RKColorGradientLBM; the initial
concentration rules (Transport2DRK.py:413-452: tracer 0 = 1 below the top buffer rows for
generated geometries, every tracer = 1 in the 10 top rows for image geometries); initial
distribution g_i = w_i C (:466-469); one tracer sub-step inside every flow step at the place the
reference does it (:1341-1418, fused into the flow kernel here); record cadence
`(iStep-1) % TimeInterval == 0`; flow datasets of SimulationResultsRK.h5 and
`/TransportMacro/TracerConcType<k>in<record>` of ConcentrationResults.h5 (:651-661).
Note the two views inside one record: the flow arrays are the state at the START of step iStep
(after streaming and boundary kernels, :1302-1312), the concentrations the state AFTER the tracer
update of that same step (:1418-1432).
"""
import warnings

import numpy as np

from . import config
from .RKD2Q9 import RKColorGradientLBM
from .results import RecordGuard, ResultFile
from .rk2d import RK2DSolver


class Transport2DRK(RKColorGradientLBM):
    def __init__(self, pathIniFile, output_dir=None, image=None, device=0, initial_dir=None, inlet_concentration_from_ini=False):
        """`inlet_concentration_from_ini`: the reference reads [BoundaryCondition] ConcentrationInlet (:162-167) and then
        hands the inlet kernel a hard-coded array [1.0] (:1161-1162, one entry: with one tracer the inlet value is 1.0
        whatever the file says; the kernel indexes it by tracer, so more than one tracer reads past its end).  Default:
        what the reference computes where it is defined (tracer 0 -> 1.0), the file's values for the others; True: the
        file's values for all."""
        RKColorGradientLBM.__init__(self, pathIniFile, output_dir=output_dir, image=image, device=device, initial_dir=initial_dir)
        self.tr = config.read_transport(pathIniFile)
        self.numTracers = self.tr["num_tracers"]
        t = self.tr
        self.inletConcentration = list(t["inlet_conc"])
        if not inlet_concentration_from_ini and t["inlet_type"] == "Dirichlet":
            if self.inletConcentration[0] != 1.0:
                warnings.warn("ConcentrationInlet[0] = %g is ignored as in the reference (Transport2DRK.py:1161 uses 1.0); "
                              "pass inlet_concentration_from_ini=True to use it" % self.inletConcentration[0])
            self.inletConcentration[0] = 1.0
        if t["outlet_type"] != "Freeflow":
            warnings.warn("OutletType = '%s': the loop applies its free-flow rows only for the spelling 'Freeflow' "
                          "(Transport2DRK.py:1363); running without a tracer outlet rule, as the reference would" % t["outlet_type"])
        if t["inlet_type"] != "Dirichlet":
            warnings.warn("InletType = '%s': the loop knows only 'Dirichlet' (Transport2DRK.py:1378); running without a "
                          "tracer inlet rule, as the reference would" % t["inlet_type"])

    def initializeTransportDomain(self):
        """Transport2DRK.py:399-469"""
        p, t = self.par, self.tr
        ny, nx = self.isDomain.shape
        rows = np.arange(ny)[:, None]
        fluid = self.isDomain == 1
        conc = np.zeros((self.numTracers, ny, nx))
        if p.get("cycle"):
            # :431-452: record LastStep of ~/LBMInitial/TransportResults.h5.  `self.tracerConc[:, :] = <record of tracer i>`
            # assigns the 2-D record to EVERY tracer on each pass of the loop over i: all tracers start from the record
            # of the last one.  (The working loop writes ConcentrationResults.h5, :651; that name is looked for second.)
            import os
            from .results import load_results
            data = None
            for stem in ("TransportResults", "ConcentrationResults"):
                for ext in (".h5", ".npz"):
                    f = os.path.join(self.initial_dir, stem + ext)
                    if data is None and os.path.isfile(f):
                        data = load_results(f)
            if data is None:
                raise config.ConfigError("IsCycle = 'yes': no TransportResults.h5/.npz (or ConcentrationResults) in %s" % self.initial_dir)
            for i in range(self.numTracers):
                key = "/TransportMacro/TracerConcType%din%d" % (i, p["last_step"])
                if key not in data:
                    raise config.ConfigError("IsCycle = 'yes': dataset %s missing" % key)
                rec = np.asarray(data[key], dtype=np.float64)
                if rec.shape != (ny, nx):
                    raise config.ConfigError("IsCycle = 'yes': %s has shape %s, the domain is %s" % (key, rec.shape, (ny, nx)))
                conc[:, :] = np.where(fluid, rec, 0.0)
        elif p["image"]:
            conc[:, (fluid & (rows >= ny - 10))] = 1.0
        else:
            conc[0, (fluid & (rows <= ny - p["nbuf"]))] = 1.0
        # [InitialCondition] TracerConc is read by the reference (:199-206) but never applied
        self.tracerConc = conc

    def runTransport2DMPMCRKNew(self, progress=None):
        p, t = self.par, self.tr
        if p["tension_type"] != "CSF":
            raise config.ConfigError("the coupled loop uses the CSF colour-gradient flow (Transport2DRK.py:1434-1485)")
        self.initializeDomainBorder()
        self.initializeDomainCondition()
        self.initializeTransportDomain()
        keys = ("sigma", "theta", "wetting", "beta", "delta", "tauR", "tauB", "tautype", "relax", "inlet", "outlet",
                "vyR", "vyB", "rhoBH", "rhoRH", "rhoBL", "rhoRL")
        solver = RK2DSolver(self.isDomain, {k: p[k] for k in keys}, device=self.device)
        self._upload_initial_state(solver)
        n = self.numTracers
        solver.configure_tracers(diffX=tuple(t["diffX"]), diffY=tuple(t["diffY"]), dXY=t["dXY"], dYX=t["dYX"],
                                 beta=(t["beta"],) * n, crit=0.5, inlet_conc=tuple(self.inletConcentration),
                                 free_outlet=t["outlet_type"] == "Freeflow", dirichlet_inlet=t["inlet_type"] == "Dirichlet", reaction_rate=t["reaction_rate"],
                                 diffJ=tuple(t["diffJ"]))
        for k in range(n):
            solver.set_tracer(k, self.tracerConc[k])
        flow = ResultFile(self.output_dir, "SimulationResultsRK",
                          (("FluidMacro", "MacroData"), ("FluidPDF", "MicroData"), ("FluidVelocity", "MacroVelocity")))
        conc = ResultFile(self.output_dir, "ConcentrationResults", (("TransportMacro", "MacroData"),))
        self.result_path, self.concentration_path = flow.path, conc.path
        self._guard = RecordGuard("rk2d+tracers", int((self.isDomain == 1).sum()), self.nan_guard)
        tracer_guard = RecordGuard("tracers", int((self.isDomain == 1).sum()), self.nan_guard)
        done = 0
        while done < self.timeSteps:
            self._step_now = done
            if done % self.timeInterval == 0:
                k = self.records
                self._record(solver, flow)              # flow view at the start of step done + 1
                solver.step(1)
                done += 1
                for i in range(n):                      # concentrations after the tracer update of that step
                    self.tracerConc[i] = solver.get_tracer(i)
                    conc.write("TransportMacro", "TracerConcType%gin%g" % (i, k), self.tracerConc[i])
                tracer_guard(k, done, {"tracer%d" % i: self.tracerConc[i] for i in range(n)},
                             {"tracer%d" % i: float(self.tracerConc[i].sum()) for i in range(n)})
            m = min(self.timeInterval - done % self.timeInterval, self.timeSteps - done) if done % self.timeInterval else 0
            if m:
                solver.step(m)
                done += m
            if progress:
                progress(done)
        solver.sync()
        self.solver = solver
        return self.result_path, self.concentration_path
