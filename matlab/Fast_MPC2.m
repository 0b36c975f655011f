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
        function U0 = first_moves(obj, X0, X0_pre, W, nw, k)
            % NOT in the reference class: the batch form of "solve, then u_prev = U(1:nu)" (README.md:555,589) for a REPLAY of many
            % timesteps / realisations with this object's model -- X0, X0_pre: n x batch, W: T*n x batch or [] -- through
            % fmpc_create + fmpc_solve_u0 (include/fastmpc.h): only the first moves (m x batch) come back over PCIe, not the
            % N_z x batch iterates (2.3 MB instead of 82 MB per 2000 problems at (27,144,30)).  Cold start (x_init is ignored);
            % nu0 = NULL (zeros) -- the dual start only enters the step-length test (SURVEY App. B-D5).
            n = size(obj.Q,1); m = size(obj.R,1); batch = size(X0,2);
            P = @(a) libpointer('doublePtr', a);
            ph = libpointer('voidPtrPtr');
            rc = calllib('libfastmpc', 'fmpc_create', ph, n, m, obj.T, 2, P(obj.A1), P(obj.A2), P(obj.B), P(obj.Q), P(obj.R), P(obj.Qf), ...
                         P(obj.q), P(obj.r), P(obj.qf), P(obj.u_min), P(obj.u_max), P(obj.x_min), P(obj.x_max), P(obj.x_final), obj.device);
            if rc < 0, error('Fast_MPC2:create', '%s', calllib('libfastmpc', 'fmpc_strerror', rc)); end
            pu = libpointer('doublePtr', zeros(m, batch));
            rc = calllib('libfastmpc', 'fmpc_solve_u0', ph.Value, batch, P(X0), P(X0_pre), P(W), P([]), P([]), nw, k, P([]), pu, ...
                         libpointer('int32Ptr', []), libpointer('int32Ptr', []));
            calllib('libfastmpc', 'fmpc_destroy', ph.Value);
            if rc < 0, error('Fast_MPC2:solve', '%s', calllib('libfastmpc', 'fmpc_strerror', rc)); end
            U0 = pu.Value;
        end
        % ---- dense builders of the reference class (VAR_2/Fast_MPC2.m:56-67).  The device path never forms H, P, C;
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
        function [P,h] = inequality_const(obj)                   % P z <= h: fast_mpc_ineq_const.m:46-56
            n = size(obj.Q,1); m = size(obj.R,1); s = n + m; T = obj.T;
            P = zeros(2*T*m, T*s); h = zeros(2*T*m,1);
            for j = 0:T-1
                iu = j*s + (1:m); r1 = 2*j*m + (1:m); r2 = (2*j+1)*m + (1:m);
                P(r1,iu) = eye(m); P(r2,iu) = -eye(m);
                h(r1) = obj.u_max; h(r2) = -obj.u_min;
            end
        end
        function [C,b] = equality_const(obj)                     % C z = b: fast_mpc_eq_const.m:38-49, terminal rows :67-71
            n = size(obj.Q,1); m = size(obj.R,1); s = n + m; T = obj.T;
            wv = obj.w; if isempty(wv), wv = zeros(T*n,1); end
            nb = T + ~isempty(obj.x_final);
            C = zeros(nb*n, T*s); b = zeros(nb*n,1);
            for i = 0:T-1
                rows = i*n + (1:n);
                C(rows, i*s + (1:m)) = -obj.B;
                C(rows, i*s + m + (1:n)) = eye(n);
                if i >= 1, C(rows, (i-1)*s + m + (1:n)) = -obj.A1; end
                if i >= 2, C(rows, (i-2)*s + m + (1:n)) = -obj.A2; end
                b(rows) = wv(i*n + (1:n));
            end
            b(1:n) = b(1:n) + obj.A1*obj.x0 + obj.A2*obj.x0_pre;          % the prediction A1 x[k-1] + A2 x[k-2]
            if T > 1, b(n+(1:n)) = b(n+(1:n)) + obj.A2*obj.x0; end
            if ~isempty(obj.x_final)
                C(T*n + (1:n), (T-1)*s + m + (1:n)) = eye(n); b(T*n + (1:n)) = obj.x_final;
            end
        end
        function [J,A_eq,b_eq] = fomulate_mpc(obj,k)             % VAR_2/Fast_MPC2.m:68-75 (name as in the reference)
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
