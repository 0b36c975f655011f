function verify_against_reference(reference_root, use_device_library)
% VERIFY_AGAINST_REFERENCE  Close the parity pin on a machine that HAS MATLAB.
%
% The build image of this repository has neither MATLAB nor Octave, so the fixtures under matlab/fixtures/ (exported from
% tests/golden/*.npz by matlab/make_fixtures.py) are outputs of the repository's own line-by-line restatement of the
% reference (oracle/dense_ref.py), NOT of the reference: "parity unpinned".  This script runs the REFERENCE's own class on
% the same inputs and the same dual start nu0 and prints the difference to the stored results, which pins the restatement
% (and with it every GPU parity test, which compares against the restatement) to the reference.
%
%   verify_against_reference('/path/to/MPC-SensorlessAO')          reference vs fixtures
%   verify_against_reference('/path/to/MPC-SensorlessAO', true)    ... and the drop-in class of this repository
%                                                                   (matlab/Fast_MPC2.m on libfastmpc.so) vs fixtures
%
% The ONLY change to the reference: inf_newton_solver.m:2 draws nu = rand(length(b),1) from the global stream.  The
% fixtures carry the nu0 that was used, so a function rand.m that returns it is put on top of the path for the duration of
% each call (a temporary directory; nothing under reference_root is touched).  Expected agreement: <= 1e-9 relative on z
% (the restatement follows the same dense algebra; differences are LAPACK/BLAS rounding).  The reference returns x_opt only,
% so iteration counts and step lengths are compared through their effect on z.
% The VAR(1) fixture (demo_var1_*) holds the INTENDED VAR(1) dynamics, i.e. the VAR_2 code with A2 = 0: the VAR_1 directory
% writes one block of C at column n instead of m + 1 (VAR_1/fast_mpc_eq_const.m:36 vs VAR_2/fast_mpc_eq_const.m:43),
% which is only right for n = m + 1; it is therefore run through VAR_2 here, box rows only.
    if nargin < 2, use_device_library = false; end
    here = fileparts(mfilename('fullpath'));
    fx = dir(fullfile(here, 'fixtures', '*.mat'));
    shadow = tempname; mkdir(shadow);
    fid = fopen(fullfile(shadow, 'rand.m'), 'w');
    fprintf(fid, 'function r = rand(varargin)\n%% returns the stored dual start of the fixture (see verify_against_reference.m)\n');
    fprintf(fid, 'global FMPC_VERIFY_NU0\nr = FMPC_VERIFY_NU0;\nassert(varargin{1} == numel(r) && varargin{2} == 1);\nend\n');
    fclose(fid);
    global FMPC_VERIFY_NU0 %#ok<GVMIS>
    worst = 0;
    for f = 1:numel(fx)
        S = load(fullfile(fx(f).folder, fx(f).name));
        n = S.n; m = S.m; T = S.T; nw = S.nw; k = S.k;
        A2 = S.A2; if S.var_order == 1, A2 = zeros(n); end
        xf = S.xf; if isempty(xf), xf = []; end
        err_ref = 0; err_dev = 0;
        for p = 1:S.num_problems
            x_init = []; if ~isempty(S.x_init), x_init = S.x_init(:, p); end
            args = {S.Q, S.R, [], S.Qf, [], [], [], S.x_min, S.x_max, S.u_min, S.u_max, [], [], T, ...
                    S.x0(:, p), S.x0_pre(:, p), zeros(m, 1), S.A1, A2, S.B, S.w(:, p), xf, x_init};
            % ---- the reference (Fast_MPC/VAR_2), rand shadowed
            FMPC_VERIFY_NU0 = S.nu0(:, p);
            addpath(fullfile(reference_root, 'Fast_MPC', 'VAR_2')); addpath(shadow);      % shadow on top
            clear Fast_MPC2 rand
            obj = Fast_MPC2(args{:});
            if nw > 0, z = obj.mpc_fixed_log_newton(nw, k); else, z = obj.mpc_fixed_log(k); end
            rmpath(shadow); rmpath(fullfile(reference_root, 'Fast_MPC', 'VAR_2'));
            clear Fast_MPC2 rand
            err_ref = max(err_ref, norm(z - S.z_expected(:, p)) / norm(S.z_expected(:, p)));
            % ---- this repository's drop-in class on the device library (draws nu0 itself: shadow rand again)
            if use_device_library
                addpath(here); addpath(shadow);
                obj = Fast_MPC2(args{:});
                if nw > 0, zd = obj.mpc_fixed_log_newton(nw, k); else, zd = obj.mpc_fixed_log(k); end
                rmpath(shadow); rmpath(here);
                clear Fast_MPC2 rand
                err_dev = max(err_dev, norm(zd - S.z_expected(:, p)) / norm(S.z_expected(:, p)));
            end
        end
        fprintf('%-34s  reference vs fixture: max rel. error on z %.2e', fx(f).name, err_ref);
        if use_device_library, fprintf('   device library vs fixture: %.2e', err_dev); end
        fprintf('\n');
        worst = max([worst, err_ref, err_dev]);
    end
    rmdir(shadow, 's');
    if worst <= 1e-9
        fprintf('PARITY PINNED: every fixture within 1e-9 of the reference (worst %.2e)\n', worst);
    else
        fprintf('MISMATCH: worst relative error %.2e (> 1e-9) -- the restatement deviates from the reference\n', worst);
    end
end
