classdef Fast_MPC2
    % Drop-in replacement for Fast_MPC/VAR_1/Fast_MPC2.m (the VAR(1) variant WITH ramp-rate rows,
    % VAR_1/fast_mpc_ineq_const.m:58-76) on the MI355X library (include/fastmpc.h, libfastmpc.so).
    % Same 21-argument constructor (VAR_1/Fast_MPC2.m:26-27) and driver methods.  var_order = 1 makes
    % fmpc_solve_once build the ramp rows du_min <= u_j - u_{j-1} <= du_max (u_{-1} = u_prev) from dumin, dumax,
    % u_prev; pass [] for any of the three to solve with the box rows only.  The device solves the intended VAR(1)
    % dynamics (the reference's misplaced row block VAR_1/fast_mpc_eq_const.m:36 is not reproduced).
    % Cannot be executed in the build image (no MATLAB); mpc-sensorlessao_amd/fast_mpc2.py (Fast_MPC2_VAR1) is the
    % tested twin.  One-time setup:  loadlibrary('libfastmpc', 'fastmpc.h')
    properties
        Q; R; S; q; r; Qf; qf; x_min; x_max; u_min; u_max; du_min; du_max
        T; x0; u_prev; A; B; w; x_final; x_init
        device = 0
    end
    methods
        function cs = Fast_MPC2(Q,R,S,Qf,q,r,qf,xmin,xmax,umin,umax,dumin,dumax,T,x0,u_prev,A,B,w,xf,x_init)
            if nargin > 1
                cs.Q = Q; cs.R = R; cs.S = S; cs.Qf = Qf; cs.q = q; cs.r = r; cs.qf = qf;
                cs.x_min = xmin; cs.x_max = xmax; cs.u_min = umin; cs.u_max = umax;
                cs.du_min = dumin; cs.du_max = dumax; cs.T = T; cs.x0 = x0; cs.u_prev = u_prev;
                cs.A = A; cs.B = B; cs.w = w; cs.x_final = xf; cs.x_init = x_init;
            end
        end
        function x_opt = mpc_fixed_log_newton(obj,nw,k)
            x_opt = obj.solve_once(obj.x_init, nw, k);
        end
        function x_opt = mpc_fixed_log(obj,k)
            x_opt = obj.solve_once(obj.x_init, 0, k);          % nw = []: <= 1000 iterations + tolerance
        end
        function x_opt = mpc_fixed_newton(obj,nw)                % VAR_1/Fast_MPC2.m, same body as VAR_2 :131-144
            x_opt = obj.k_schedule(nw);
        end
        function x_opt = mpc_solve_full(obj)                     % same body as VAR_2 :100-115
            x_opt = obj.k_schedule(0);
        end
        function x_opt = mpc_solve_check(obj,k_min,k_max)        % same body as VAR_2 :88-99
            ks = linspace(k_max,k_min,5); z = obj.initialize();
            for i = 1:numel(ks), z = obj.solve_once(z, 0, ks(i)); end
            x_opt = z;
        end
        % ---- dense builders of the reference class (VAR_1/Fast_MPC2.m:52-63).  The device path never forms H, P, C;
        % these are host-side MATLAB, written from the index maps of the solver (z = [u0;x1;u1;x2;...;u_{T-1};x_T]) and
        % kept so that callers of objective_function / inequality_const / equality_const / fomulate_mpc keep working.
        % Python twin (tested against the dense restatement): mpc-sensorlessao_amd/fast_mpc2.py.
        function [H,g] = objective_function(obj)                 % cost z'Hz + g'z (no 1/2): fast_mpc_objective.m:50-65
            n = size(obj.Q,1); m = size(obj.R,1); s = n + m; T = obj.T;
            H = zeros(T*s); g = zeros(T*s,1);
            qv = obj.q; if isempty(qv), qv = zeros(n,1); end
            rv = obj.r; if isempty(rv), rv = zeros(m,1); end
            qfv = obj.qf; if isempty(qfv), qfv = zeros(n,1); end
            for j = 0:T-1
                iu = j*s + (1:m); ix = j*s + m + (1:n);
                H(iu,iu) = obj.R; g(iu) = rv;
                if j == T-1, H(ix,ix) = obj.Qf; g(ix) = qfv; else, H(ix,ix) = obj.Q; g(ix) = qv; end
            end
        end
        function [P,h] = inequality_const(obj)                   % box rows, then ramp rows: VAR_1/fast_mpc_ineq_const.m:42-79
            n = size(obj.Q,1); m = size(obj.R,1); s = n + m; T = obj.T;
            P = zeros(2*T*m, T*s); h = zeros(2*T*m,1);
            for j = 0:T-1
                iu = j*s + (1:m); r1 = 2*j*m + (1:m); r2 = (2*j+1)*m + (1:m);
                P(r1,iu) = eye(m); P(r2,iu) = -eye(m);
                h(r1) = obj.u_max; h(r2) = -obj.u_min;
            end
            if isempty(obj.du_min) || isempty(obj.du_max) || isempty(obj.u_prev), return; end   % box rows only
            Pr = zeros(2*T*m, T*s); hr = zeros(2*T*m,1);         % du_min <= u_j - u_{j-1} <= du_max, u_{-1} = u_prev
            for j = 0:T-1
                iu = j*s + (1:m); r1 = 2*j*m + (1:m); r2 = (2*j+1)*m + (1:m);
                Pr(r1,iu) = eye(m); Pr(r2,iu) = -eye(m);
                up = zeros(m,1);
                if j >= 1, ip = (j-1)*s + (1:m); Pr(r1,ip) = -eye(m); Pr(r2,ip) = eye(m); else, up = obj.u_prev; end
                hr(r1) = up + obj.du_max; hr(r2) = -up - obj.du_min;
            end
            P = [P; Pr]; h = [h; hr];
        end
        function [C,b] = equality_const(obj)                     % C z = b: VAR_1/fast_mpc_eq_const.m:32-55
            n = size(obj.Q,1); m = size(obj.R,1); s = n + m; T = obj.T;
            wv = obj.w; if isempty(wv), wv = zeros(T*n,1); end
            nb = T + ~isempty(obj.x_final);
            C = zeros(nb*n, T*s); b = zeros(nb*n,1);
            for i = 0:T-1
                rows = i*n + (1:n);
                C(rows, i*s + (1:m)) = -obj.B;
                C(rows, i*s + m + (1:n)) = eye(n);
                if i >= 1, C(rows, (i-1)*s + m + (1:n)) = -obj.A; end     % intended VAR(1) dynamics (the reference writes this
                                                                            % block at column n instead of m+1 for i = 1: VAR_1/fast_mpc_eq_const.m:36)
                b(rows) = wv(i*n + (1:n));
            end
            b(1:n) = b(1:n) + obj.A*obj.x0;                               % the prediction A x[k-1]
            if ~isempty(obj.x_final)
                C(T*n + (1:n), (T-1)*s + m + (1:n)) = eye(n); b(T*n + (1:n)) = obj.x_final;
            end
        end
        function [J,A_eq,b_eq] = fomulate_mpc(obj,k)             % VAR_1/Fast_MPC2.m:64-71 (name as in the reference)
            [P,h] = obj.inequality_const(); [H,g] = obj.objective_function();
            J = @(z)(z'*H*z + g'*z + k*(-sum(log(h - P*z))));
            [A_eq,b_eq] = obj.equality_const();
        end
        function z_init = initialize(obj)                        % fast_mpc_init.m:12-27
            n = size(obj.Q,1); m = size(obj.R,1);
            if ~isempty(obj.x_init), z_init = obj.x_init; return; end
            z_init = repmat([(obj.u_min+obj.u_max)/2; (obj.x_min+obj.x_max)/2], obj.T, 1);
            assert(numel(z_init) == obj.T*(n+m));
        end
    end
    methods (Access = private)
        function x_opt = k_schedule(obj,nw)                      % k = 1, x0.1 while k*length(z) >= 10e-3, warm starts
            k = 1; mu = 1/10; z = obj.initialize(); x_opt = z;
            while k*length(z) >= 10e-3
                x_opt = obj.solve_once(z, nw, k); k = mu*k; z = x_opt;
            end
        end
        function x_opt = solve_once(obj, z_init, nw, k)
            n = size(obj.Q,1); m = size(obj.R,1); Nz = obj.T*(n+m);
            nu0 = rand(obj.T*n + n*(~isempty(obj.x_final)), 1);  % inf_newton_solver.m:2, drawn here
            P = @(a) libpointer('doublePtr', a);                 % [] -> NULL
            % outputs are read back from libpointers kept in variables (calllib writes into the pointer's own buffer)
            pz = libpointer('doublePtr', zeros(Nz,1));
            pit = libpointer('int32Ptr', int32(0));
            rc = calllib('libfastmpc','fmpc_solve_once', n, m, obj.T, 1, ...
                P(obj.Q),P(obj.R),P(obj.S),P(obj.Qf),P(obj.q),P(obj.r),P(obj.qf), ...
                P(obj.x_min),P(obj.x_max),P(obj.u_min),P(obj.u_max),P(obj.du_min),P(obj.du_max), ...
                P(obj.x0),P([]),P(obj.u_prev),P(obj.A),P([]),P(obj.B), ...
                P(obj.w),P(obj.x_final),P(z_init),P(nu0), int32(nw), k, int32(obj.device), ...
                pz, pit);
            if rc < 0, error('fastmpc:%d %s', rc, calllib('libfastmpc','fmpc_strerror',rc)); end
            x_opt = pz.Value;
        end
    end
end
