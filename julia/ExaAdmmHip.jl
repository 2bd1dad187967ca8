# ExaAdmmHip.jl -- binding of libexaadmm_hip.so (include/exaadmm_hip.h) into ExaAdmm.jl.
#
# UNEXECUTED SOURCE: no Julia runtime exists in the build image; every entry point used here
# is exercised from Python/ctypes by tests/ through the same C ABI.
#
# Usage inside ExaAdmm (after `include("ExaAdmmHip.jl")` at the end of src/ExaAdmm.jl):
#     env, mod = ExaAdmm.solve_acopf_hip("case13659pegase.m"; rho_pq=5e1, rho_va=5e3, scale=1e-4,
#                                        tight_factor=0.99)
# The reference's dispatch seam is the third positional argument `device` of the generic
# functions called by admm_two_level (src/algorithms/admm_two_level.jl:29-81); `HipNative`
# is a new concrete type there, so no existing method is touched.

const libexaadmm_hip = get(ENV, "EXAADMM_HIP_LIB", "libexaadmm_hip.so")

struct ExaadmmGrid            # == exaadmm_grid
    ngen::Int64; nline::Int64; nbus::Int64
    index_base::Int32
    baseMVA::Float64
    pgmin::Ptr{Float64}; pgmax::Ptr{Float64}; qgmin::Ptr{Float64}; qgmax::Ptr{Float64}
    c2::Ptr{Float64}; c1::Ptr{Float64}; c0::Ptr{Float64}
    YffR::Ptr{Float64}; YffI::Ptr{Float64}; YftR::Ptr{Float64}; YftI::Ptr{Float64}
    YttR::Ptr{Float64}; YttI::Ptr{Float64}; YtfR::Ptr{Float64}; YtfI::Ptr{Float64}
    FrVmBound::Ptr{Float64}; ToVmBound::Ptr{Float64}; FrVaBound::Ptr{Float64}; ToVaBound::Ptr{Float64}
    rateA::Ptr{Float64}
    brBusIdx::Ptr{Int64}
    Pd::Ptr{Float64}; Qd::Ptr{Float64}; Vmin::Ptr{Float64}; Vmax::Ptr{Float64}
    YshR::Ptr{Float64}; YshI::Ptr{Float64}
    FrStart::Ptr{Int64}; FrIdx::Ptr{Int64}; ToStart::Ptr{Int64}; ToIdx::Ptr{Int64}
    GenStart::Ptr{Int64}; GenIdx::Ptr{Int64}
    nline_owned::Int64   # partitions only: branches past it are replicas of a peer's cut branches; 0 = all (single GPU)
end

mutable struct ExaadmmParams  # == exaadmm_params
    mu_max::Float64; max_auglag::Int32; initial_beta::Float64; beta::Float64; inc_c::Float64
    theta::Float64; outer_eps::Float64; MAX_MULTIPLIER::Float64; outer_iterlim::Int32
    inner_iterlim::Int32; scale::Float64; verbose::Int32; check_every::Int32
end

"Device tag: owns the opaque library handle of one MI355X."
mutable struct HipNative
    handle::Ptr{Cvoid}
end

const U_CURR, V_CURR, L_CURR, RHO, RD, RP, Z_CURR, Z_PREV, LZ, AX_PLUS_BY = 0:9

function hipcheck(rc::Cint, what)
    rc == 0 || error("$what failed ($rc): " * unsafe_string(ccall((:exaadmm_last_error, libexaadmm_hip), Cstring, ())))
    return nothing
end

to_params(par::Parameters) = ExaadmmParams(par.mu_max, par.max_auglag, par.initial_beta, par.beta, par.inc_c,
    par.theta, par.outer_eps, par.MAX_MULTIPLIER, par.outer_iterlim, par.inner_iterlim, par.scale,
    par.verbose, 1)

"Create the device model from a CPU-typed ModelAcopf (arrays stay 1-based: index_base = 1)."
function HipNative(env::AdmmEnv, mod::ModelAcopf; gpu_no::Int=0)
    d = mod.grid_data
    g = ExaadmmGrid(d.ngen, d.nline, d.nbus, 1, d.baseMVA,
        pointer(d.pgmin), pointer(d.pgmax), pointer(d.qgmin), pointer(d.qgmax), pointer(d.c2), pointer(d.c1), pointer(d.c0),
        pointer(d.YffR), pointer(d.YffI), pointer(d.YftR), pointer(d.YftI), pointer(d.YttR), pointer(d.YttI),
        pointer(d.YtfR), pointer(d.YtfI), pointer(d.FrVmBound), pointer(d.ToVmBound), pointer(d.FrVaBound),
        pointer(d.ToVaBound), pointer(d.rateA), pointer(d.brBusIdx), pointer(d.Pd), pointer(d.Qd), pointer(d.Vmin),
        pointer(d.Vmax), pointer(d.YshR), pointer(d.YshI), pointer(d.FrStart), pointer(d.FrIdx), pointer(d.ToStart),
        pointer(d.ToIdx), pointer(d.GenStart), pointer(d.GenIdx), 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    p = to_params(env.params)
    GC.@preserve d begin
        hipcheck(ccall((:exaadmm_create, libexaadmm_hip), Cint, (Ref{ExaadmmGrid}, Ref{ExaadmmParams}, Cint, Ref{Ptr{Cvoid}}),
                       g, p, gpu_no, h), "exaadmm_create")
    end
    dev = HipNative(h[])
    finalizer(x -> ccall((:exaadmm_destroy, libexaadmm_hip), Cint, (Ptr{Cvoid},), x.handle), dev)
    upload!(dev, mod)
    return dev
end

function upload!(dev::HipNative, mod::ModelAcopf)
    s = mod.solution; n = length(s.u_curr)
    for (id, v) in ((U_CURR, s.u_curr), (V_CURR, s.v_curr), (L_CURR, s.l_curr), (RHO, s.rho), (Z_CURR, s.z_curr),
                    (Z_PREV, s.z_prev), (LZ, s.lz))
        hipcheck(ccall((:exaadmm_upload_solution, libexaadmm_hip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Int64),
                       dev.handle, id, v, n), "exaadmm_upload_solution")
    end
    hipcheck(ccall((:exaadmm_upload_membuf, libexaadmm_hip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64),
                   dev.handle, mod.membuf, 31, size(mod.membuf, 2)), "exaadmm_upload_membuf")
    hipcheck(ccall((:exaadmm_set_pg_bounds, libexaadmm_hip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64),
                   dev.handle, mod.pgmin_curr, mod.pgmax_curr, length(mod.pgmin_curr)), "exaadmm_set_pg_bounds")
end

function download!(dev::HipNative, mod::ModelAcopf)
    s = mod.solution; n = length(s.u_curr)
    for (id, v) in ((U_CURR, s.u_curr), (V_CURR, s.v_curr), (L_CURR, s.l_curr), (RD, s.rd), (RP, s.rp),
                    (Z_CURR, s.z_curr), (Z_PREV, s.z_prev), (LZ, s.lz), (AX_PLUS_BY, s.Ax_plus_By))
        hipcheck(ccall((:exaadmm_download_solution, libexaadmm_hip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Int64),
                       dev.handle, id, v, n), "exaadmm_download_solution")
    end
    hipcheck(ccall((:exaadmm_download_membuf, libexaadmm_hip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64),
                   dev.handle, mod.membuf, 31, size(mod.membuf, 2)), "exaadmm_download_membuf")
end

# ---- options and diagnostics of a handle (include/exaadmm_hip.h: the library reads no environment variable of its own)
function set_option(dev::HipNative, key::AbstractString, value::Integer)
    hipcheck(ccall((:exaadmm_set_option, libexaadmm_hip), Cint, (Ptr{Cvoid}, Cstring, Int64), dev.handle, key, value), "exaadmm_set_option($key)")
    return
end
function get_option(dev::HipNative, key::AbstractString)
    r = Ref{Int64}(0)
    hipcheck(ccall((:exaadmm_get_option, libexaadmm_hip), Cint, (Ptr{Cvoid}, Cstring, Ref{Int64}), dev.handle, key, r), "exaadmm_get_option($key)")
    return r[]
end
# which loop the fused entry points use (0 four launches per iteration, 1 two, 2 resident) and what the last call launched
function loop_kind(dev::HipNative)
    r = Ref{Cint}(0)
    hipcheck(ccall((:exaadmm_loop_kind, libexaadmm_hip), Cint, (Ptr{Cvoid}, Ref{Cint}), dev.handle, r), "exaadmm_loop_kind")
    return Int(r[])
end
function resident_form(dev::HipNative)
    t, hv, n, w = Ref{Cint}(0), Ref{Cint}(0), Ref{Cint}(0), Ref{Cint}(0)
    hipcheck(ccall((:exaadmm_resident_form, libexaadmm_hip), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ref{Cint}, Ref{Cint}), dev.handle, t, hv, n, w), "exaadmm_resident_form")
    return (tiles = Int(t[]), heavy = Int(hv[]), launches = Int(n[]), wavefronts = Int(w[]))
end
function resident_selfchecks(dev::HipNative)
    a, b = Ref{Cint}(0), Ref{Cint}(0)
    hipcheck(ccall((:exaadmm_resident_selfcheck_stats, libexaadmm_hip), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}), dev.handle, a, b), "exaadmm_resident_selfcheck_stats")
    return (checks = Int(a[]), mismatches = Int(b[]))
end

# ---- the plugin steps (docs/src/dev.md:65-155); counters keep the generic untyped methods of
# ---- acopf_admm_increment.jl, which only touch mod.info
function admm_outer_prestep(env::AdmmEnv, mod::AbstractOPFModel, dev::HipNative)
    r = Ref{Float64}(0.0)
    hipcheck(ccall((:exaadmm_norm_z, libexaadmm_hip), Cint, (Ptr{Cvoid}, Ref{Float64}), dev.handle, r), "exaadmm_norm_z")
    mod.info.norm_z_prev = r[]
    return
end
admm_inner_prestep(env::AdmmEnv, mod::AbstractOPFModel, dev::HipNative) =
    hipcheck(ccall((:exaadmm_inner_prestep, libexaadmm_hip), Cint, (Ptr{Cvoid},), dev.handle), "exaadmm_inner_prestep")
admm_update_x(env::AdmmEnv, mod::AbstractOPFModel, dev::HipNative) =
    hipcheck(ccall((:exaadmm_update_x, libexaadmm_hip), Cint, (Ptr{Cvoid}, Cint), dev.handle, mod.info.inner), "exaadmm_update_x")
admm_update_xbar(env::AdmmEnv, mod::AbstractOPFModel, dev::HipNative) =
    hipcheck(ccall((:exaadmm_update_xbar, libexaadmm_hip), Cint, (Ptr{Cvoid},), dev.handle), "exaadmm_update_xbar")
admm_update_z(env::AdmmEnv, mod::AbstractOPFModel, dev::HipNative) =
    hipcheck(ccall((:exaadmm_update_z, libexaadmm_hip), Cint, (Ptr{Cvoid}, Cdouble), dev.handle, env.params.beta), "exaadmm_update_z")
admm_update_l(env::AdmmEnv, mod::AbstractOPFModel, dev::HipNative) =
    hipcheck(ccall((:exaadmm_update_l, libexaadmm_hip), Cint, (Ptr{Cvoid}, Cdouble), dev.handle, env.params.beta), "exaadmm_update_l")
function admm_update_residual(env::AdmmEnv, mod::AbstractOPFModel, dev::HipNative)
    out = zeros(4)
    hipcheck(ccall((:exaadmm_update_residual, libexaadmm_hip), Cint, (Ptr{Cvoid}, Ptr{Float64}), dev.handle, out), "exaadmm_update_residual")
    mod.info.primres, mod.info.dualres, mod.info.norm_z_curr, mod.info.mismatch = out
    return
end
admm_update_lz(env::AdmmEnv, mod::AbstractOPFModel, dev::HipNative) =
    hipcheck(ccall((:exaadmm_update_lz, libexaadmm_hip), Cint, (Ptr{Cvoid}, Cdouble), dev.handle, env.params.beta), "exaadmm_update_lz")
function admm_poststep(env::AdmmEnv, mod::AbstractOPFModel, dev::HipNative)
    r = Ref{Float64}(0.0)
    hipcheck(ccall((:exaadmm_objective, libexaadmm_hip), Cint, (Ptr{Cvoid}, Ref{Float64}), dev.handle, r), "exaadmm_objective")
    mod.info.objval = r[]
    download!(dev, mod)
    return
end

"solve_acopf with the MI355X backend; same keyword surface as interface/solve_acopf.jl:1-7."
function solve_acopf_hip(case::String; case_format="matpower", outer_iterlim=20, inner_iterlim=1000,
        rho_pq=400.0, rho_va=40000.0, obj_scale=1.0, scale=1e-4, tight_factor=1.0, outer_eps=2e-4,
        gpu_no=0, verbose=1, fused=false, options=Dict{String,Int}())
    # options: A/B switches of the device model, e.g. Dict("resident" => 0) (include/exaadmm_hip.h exaadmm_set_option; none is needed)
    T = Float64; TD = Array{Float64,1}; TI = Array{Int,1}; TM = Array{Float64,2}
    env = AdmmEnv{T,TD,TI,TM}(case, rho_pq, rho_va; case_format=case_format, use_gpu=false,
                              tight_factor=tight_factor, verbose=verbose)
    mod = ModelAcopf{T,TD,TI,TM}(env)
    env.params.scale = scale; env.params.obj_scale = obj_scale; env.params.outer_eps = outer_eps
    env.params.outer_iterlim = outer_iterlim; env.params.inner_iterlim = inner_iterlim
    dev = HipNative(env, mod; gpu_no=gpu_no)
    for (key, value) in options
        set_option(dev, key, value)
    end
    if fused   # whole driver inside the library, loop control on the device
        info = zeros(UInt8, 160)
        hipcheck(ccall((:exaadmm_solve_two_level, libexaadmm_hip), Cint, (Ptr{Cvoid}, Ptr{UInt8}), dev.handle, info),
                 "exaadmm_solve_two_level")
        ints = reinterpret(Int32, info[1:16]); dbl = reinterpret(Float64, info[17:end])
        mod.info.status = ints[1] == 1 ? :Solved : :IterationLimit
        mod.info.inner, mod.info.outer, mod.info.cumul = ints[2], ints[3], ints[4]
        mod.info.objval, mod.info.primres, mod.info.dualres, mod.info.mismatch = dbl[1:4]
        mod.info.norm_z_curr, mod.info.norm_z_prev, mod.info.eps_pri, mod.info.time_overall = dbl[5:8]
        download!(dev, mod)
    else       # the reference's own driver, unchanged, calling the step-wise entry points
        admm_two_level(env, mod, dev)
    end
    return env, mod
end

# ---------------------------------------------------------------------------------------------------
# Multi-period ACOPF (ModelMpacopf, models/mpacopf/mpacopf_model.jl:40-112): ONE time-expanded device
# model (exaadmm_mp_create).  The reference's generic driver runs unchanged: every admm_* method above
# is typed on AbstractOPFModel and therefore also serves ModelMpacopf with a HipNative device -- the
# library applies the multi-period semantics (ramp subproblems, ramp terms of the bus update, ramp
# blocks of z/l/lz, max-over-periods residual norms, eps_pri from mod.nvar) behind the same entry points.
# Counters: mpacopf_admm_increment.jl types its methods `device::Nothing`, so with a HipNative device the
# 3-argument calls of admm_two_level fall through to the generic AbstractOPFModel methods
# (acopf_admm_increment.jl), which update mod.info -- all the library needs.
"Device model of a CPU-typed ModelMpacopf; loads come from the periods' grid_data (mpacopf_model.jl:77-80)."
function HipNative(env::AdmmEnv, mod::ModelMpacopf; gpu_no::Int=0)
    d = mod.models[1].grid_data
    T = mod.len_horizon
    pd = reduce(vcat, [mod.models[i].grid_data.Pd for i in 1:T])   # T x nbus, period-major
    qd = reduce(vcat, [mod.models[i].grid_data.Qd for i in 1:T])
    g = ExaadmmGrid(d.ngen, d.nline, d.nbus, 1, d.baseMVA,
        pointer(d.pgmin), pointer(d.pgmax), pointer(d.qgmin), pointer(d.qgmax), pointer(d.c2), pointer(d.c1), pointer(d.c0),
        pointer(d.YffR), pointer(d.YffI), pointer(d.YftR), pointer(d.YftI), pointer(d.YttR), pointer(d.YttI),
        pointer(d.YtfR), pointer(d.YtfI), pointer(d.FrVmBound), pointer(d.ToVmBound), pointer(d.FrVaBound),
        pointer(d.ToVaBound), pointer(d.rateA), pointer(d.brBusIdx), pointer(d.Pd), pointer(d.Qd), pointer(d.Vmin),
        pointer(d.Vmax), pointer(d.YshR), pointer(d.YshI), pointer(d.FrStart), pointer(d.FrIdx), pointer(d.ToStart),
        pointer(d.ToIdx), pointer(d.GenStart), pointer(d.GenIdx), 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    p = to_params(env.params)
    GC.@preserve d pd qd begin
        hipcheck(ccall((:exaadmm_mp_create, libexaadmm_hip), Cint,
                       (Ref{ExaadmmGrid}, Ref{ExaadmmParams}, Cint, Ptr{Float64}, Ptr{Float64}, Cdouble, Cint, Ref{Ptr{Cvoid}}),
                       g, p, T, pd, qd, mod.ramp_ratio, gpu_no, h), "exaadmm_mp_create")
    end
    dev = HipNative(h[])
    finalizer(x -> ccall((:exaadmm_destroy, libexaadmm_hip), Cint, (Ptr{Cvoid},), x.handle), dev)
    hipcheck(ccall((:exaadmm_init_solution, libexaadmm_hip), Cint, (Ptr{Cvoid}, Cdouble, Cdouble),
                   dev.handle, env.initial_rho_pq, env.initial_rho_va), "exaadmm_init_solution")
    return dev
end

"Copy the device state back into mod.models[i].solution / mod.solution[i] (reference layouts)."
function download!(dev::HipNative, mod::ModelMpacopf)
    T = mod.len_horizon; G = mod.models[1].grid_data.ngen; L = mod.models[1].grid_data.nline
    n = T * (2G + 8L)
    buf = zeros(n); rbuf = zeros(T * G)
    for (id, f) in ((0, :u_curr), (1, :v_curr), (2, :l_curr), (3, :rho), (6, :z_curr), (7, :z_prev), (8, :lz))
        hipcheck(ccall((:exaadmm_download_solution, libexaadmm_hip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Int64),
                       dev.handle, id, buf, n), "exaadmm_download_solution")
        for t in 1:T   # expanded layout: [gens of all periods | lines of all periods]
            v = getfield(mod.models[t].solution, f)
            v[1:2G] .= buf[2G*(t-1)+1:2G*t]
            v[2G+1:2G+8L] .= buf[2G*T+8L*(t-1)+1:2G*T+8L*t]
        end
        if f != :v_curr
            hipcheck(ccall((:exaadmm_mp_download_ramp, libexaadmm_hip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Int64),
                           dev.handle, id, rbuf, T * G), "exaadmm_mp_download_ramp")
            for t in 2:T
                getfield(mod.solution[t], f) .= rbuf[G*(t-1)+1:G*t]
            end
        end
    end
    hipcheck(ccall((:exaadmm_mp_download_ramp, libexaadmm_hip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Int64),
                   dev.handle, 20, rbuf, T * G), "exaadmm_mp_download_ramp")   # s_curr
    for t in 2:T
        mod.solution[t].s_curr .= rbuf[G*(t-1)+1:G*t]
    end
end

"solve_mpacopf with the MI355X backend; keyword surface of interface/solve_mpacopf.jl:1-6 (warm_start=false)."
function solve_mpacopf_hip(case::String, load_prefix::String; case_format="matpower", start_period=1, end_period=1,
        outer_iterlim=20, inner_iterlim=1000, rho_pq=400.0, rho_va=40000.0, obj_scale=1.0, scale=1e-4,
        tight_factor=1.0, outer_eps=2e-4, gpu_no=0, verbose=1, ramp_ratio=0.02)
    T = Float64; TD = Array{Float64,1}; TI = Array{Int,1}; TM = Array{Float64,2}
    env = AdmmEnv{T,TD,TI,TM}(case, rho_pq, rho_va; case_format=case_format, use_gpu=false,
                              load_prefix=load_prefix, tight_factor=tight_factor, verbose=verbose)
    mod = ModelMpacopf{T,TD,TI,TM}(env; start_period=start_period, end_period=end_period, ramp_ratio=ramp_ratio)
    env.params.scale = scale; env.params.obj_scale = obj_scale; env.params.outer_eps = outer_eps
    env.params.outer_iterlim = outer_iterlim; env.params.inner_iterlim = inner_iterlim
    dev = HipNative(env, mod; gpu_no=gpu_no)
    admm_two_level(env, mod, dev)      # the reference's driver; admm_poststep downloads
    return env, mod
end
