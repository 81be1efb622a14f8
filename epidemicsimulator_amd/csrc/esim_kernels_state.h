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

