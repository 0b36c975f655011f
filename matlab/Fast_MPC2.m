classdef Fast_MPC2
    % Drop-in replacement for Fast_MPC/VAR_2/Fast_MPC2.m of jinsungkim96/MPC-SensorlessAO that
    % sends every solve through the MI355X library (include/fastmpc.h, libfastmpc.so).
    % Same constructor (23 arguments) and the same driver methods; x_opt is the interleaved
    % z = [u0;x1;u1;x2;...].  Written against the C ABI; it cannot be executed in the build image
    % (no MATLAB) -- the Python mirror mpc-sensorlessao_amd/fast_mpc2.py is the tested twin.
    %
    % One-time setup:  loadlibrary('libfastmpc', 'fastmpc.h')
    properties
        Q; R; S; q; r; Qf; qf; x_min; x_max; u_min; u_max; du_min; du_max
        T; x0; x0_pre; u_prev; A1; A2; B; w; x_final; x_init
        device = 0
    end
    methods
        function cs = Fast_MPC2(Q,R,S,Qf,q,r,qf,xmin,xmax,umin,umax,dumin,dumax,T,x0,x0_pre,u_prev,...
                A1,A2,B,w,xf,x_init)
            if nargin > 1
                cs.Q = Q; cs.R = R; cs.S = S; cs.Qf = Qf; cs.q = q; cs.r = r; cs.qf = qf;
                cs.x_min = xmin; cs.x_max = xmax; cs.u_min = umin; cs.u_max = umax;
                cs.du_min = dumin; cs.du_max = dumax; cs.T = T; cs.x0 = x0; cs.x0_pre = x0_pre;
                cs.u_prev = u_prev; cs.A1 = A1; cs.A2 = A2; cs.B = B; cs.w = w; cs.x_final = xf;
                cs.x_init = x_init;
            end
        end
        function x_opt = mpc_fixed_log_newton(obj,nw,k)
            x_opt = obj.solve_once(obj.x_init, nw, k);
        end
        function x_opt = mpc_fixed_log(obj,k)
            x_opt = obj.solve_once(obj.x_init, 0, k);          % nw = []: <= 1000 iterations + tolerance
        end
        function x_opt = mpc_fixed_newton(obj,nw)
            x_opt = obj.k_schedule(nw);
        end
        function x_opt = mpc_solve_full(obj)
            x_opt = obj.k_schedule(0);
        end
        function x_opt = mpc_solve_check(obj,k_min,k_max)
            ks = linspace(k_max,k_min,5); z = obj.initialize();
            for i = 1:numel(ks), z = obj.solve_once(z, 0, ks(i)); end
            x_opt = z;
        end
        function z_init = initialize(obj)                        % fast_mpc_init.m:12-27
            n = size(obj.Q,1); m = size(obj.R,1);
            if ~isempty(obj.x_init), z_init = obj.x_init; return; end
            z_init = repmat([(obj.u_min+obj.u_max)/2; (obj.x_min+obj.x_max)/2], obj.T, 1);
            assert(numel(z_init) == obj.T*(n+m));
        end
    end
    methods (Access = private)
        function x_opt = k_schedule(obj,nw)                      % Fast_MPC2.m:100-115,131-144
            k = 1; mu = 1/10; z = obj.initialize(); x_opt = z;
            while k*length(z) >= 10e-3
                x_opt = obj.solve_once(z, nw, k); k = mu*k; z = x_opt;
            end
        end
        function x_opt = solve_once(obj, z_init, nw, k)
            n = size(obj.Q,1); m = size(obj.R,1); Nz = obj.T*(n+m);
            nu0 = rand(obj.T*n + n*(~isempty(obj.x_final)), 1);  % inf_newton_solver.m:2, drawn here
            P = @(a) libpointer('doublePtr', a);                 % [] -> NULL
            % calllib copies INTO a libpointer's own buffer: the outputs must be read back from pointers that
            % are kept in variables (a temporary libpointer, or a plain array argument, would be lost)
            pz = libpointer('doublePtr', zeros(Nz,1));
            pit = libpointer('int32Ptr', int32(0));
            rc = calllib('libfastmpc','fmpc_solve_once', n, m, obj.T, 2, ...
                P(obj.Q),P(obj.R),P(obj.S),P(obj.Qf),P(obj.q),P(obj.r),P(obj.qf), ...
                P(obj.x_min),P(obj.x_max),P(obj.u_min),P(obj.u_max),P(obj.du_min),P(obj.du_max), ...
                P(obj.x0),P(obj.x0_pre),P(obj.u_prev),P(obj.A1),P(obj.A2),P(obj.B), ...
                P(obj.w),P(obj.x_final),P(z_init),P(nu0), int32(nw), k, int32(obj.device), ...
                pz, pit);
            if rc < 0, error('fastmpc:%d %s', rc, calllib('libfastmpc','fmpc_strerror',rc)); end
            x_opt = pz.Value;                                    % N_z x 1, interleaved [u0;x1;u1;x2;...]
        end
    end
end
