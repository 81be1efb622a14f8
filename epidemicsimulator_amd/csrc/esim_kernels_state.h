// esim_kernels_state.h -- read-back: per-citizen state in the reference's terms, the exposure log.
#pragma once
__global__ __launch_bounds__(TPB) void k_export_log(Dev d, uint32_t first, uint32_t n, uint32_t *citizen, uint8_t *on_bus)
{
    for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
        const uint32_t c = d.log[first + i];
        citizen[i] = c;
        on_bus[i] = (d.cit[c] & CW_BUS_EXPOSED) ? 1 : 0;
    }
}

__global__ __launch_bounds__(TPB) void k_decode_state(Dev d, uint8_t *status, uint16_t *timer, uint32_t *cur,
                                                      uint8_t *on_bus, uint8_t *elig)
{
    const Ctrl *ctrl = d.ctrl;
    const uint32_t t = ctrl->t - 1u;             // last completed step
    for (uint32_t c = blockIdx.x * TPB + threadIdx.x; c < d.n; c += gridDim.x * TPB) {
        const uint32_t st = d.cit[c], te = CW_TE(st), fl = st & CW_FLAGS;
        const uint32_t cls = status_of(te, t, d.exposed_time, d.infected_time);
        uint32_t tm = 0;
        if (te < TE_RECOVERED) {
            const uint32_t dd = t + TE_BIAS - te;
            if (cls == ESIM_EXPOSED) tm = dd; else if (cls == ESIM_INFECTED) tm = dd - d.exposed_time - 1u;
        }
        if (status) status[c] = (uint8_t)cls;
        if (timer) timer[c] = (uint16_t)tm;
        if (cur) cur[c] = (ctrl->at_work && (fl & FL_HAS_WORK)) ? d.work[c] : d.home[c];
        if (on_bus) on_bus[c] = (fl & FL_USES_PT) ? (uint8_t)ctrl->bus_dir : 0;
        if (elig) elig[c] = (ctrl->have_elig && eligible(st, ctrl->trigger_step)) ? 1 : 0;
    }
}


// Sharded runs: steps drawn as chunks hold this shard's census; the additive fields of records [first, first + n) are summed
// over the shards (buffer R) and written back, disease_exists follows.
#define XR_FIELDS 9u
__global__ __launch_bounds__(TPB) void k_records_pack(Dev d, uint32_t first, uint32_t n, uint32_t *xr)
{
    for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
        const esim_step_result r = d.records[first + i];
        uint32_t *o = xr + (size_t)i * XR_FIELDS;
        o[0] = r.susceptible; o[1] = r.exposed; o[2] = r.infected; o[3] = r.recovered; o[4] = r.vaccinated;
        o[5] = r.exposures_building; o[6] = r.exposures_bus; o[7] = r.n_riders; o[8] = r.eligible_count;
    }
}

__global__ __launch_bounds__(TPB) void k_records_unpack(Dev d, uint32_t first, uint32_t n, const uint32_t *xr)
{
    for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
        esim_step_result r = d.records[first + i];
        const uint32_t *o = xr + (size_t)i * XR_FIELDS;
        r.susceptible = o[0]; r.exposed = o[1]; r.infected = o[2]; r.recovered = o[3]; r.vaccinated = o[4];
        r.exposures_building = o[5]; r.exposures_bus = o[6]; r.n_riders = o[7]; r.eligible_count = o[8];
        r.disease_exists = (r.exposed != 0u || r.infected != 0u || r.susceptible != 0u) ? 1u : 0u;
        d.records[first + i] = r;
    }
}

// Sharded runs: every host read-back of the control block is preceded by a SUM all-reduce of the shards' error fields, so that
// every rank sees any rank's device-side error in the same collective and all leave esim_run_sharded together with the
// same code (a rank that left alone would leave its peers inside their next collective).
__global__ __launch_bounds__(64) void k_status_pack(Dev d)
{
    if (threadIdx.x == 0) { d.xe[0] = ERR_FIELD(d.ctrl->error); d.xe[1] = d.ctrl->finished ? 1u : 0u; }
    if (threadIdx.x < ERR_MAX_WORLD) d.xe[2u + threadIdx.x] = threadIdx.x == d.rank ? d.ctrl->xs_need : 0u;
}

__global__ __launch_bounds__(64) void k_status_unpack(Dev d)
{
    if (threadIdx.x == 0) {
        d.ctrl->peer_error |= d.xe[0]; if (d.xe[0] && !d.ctrl->error) d.ctrl->error = err_decode(d.xe[0]);
        uint32_t need = 0u;
        for (uint32_t r = 0; r < ERR_MAX_WORLD; ++r) need = max(need, d.xe[2u + r]);
        d.ctrl->xs_need_all = need;
    }
}
