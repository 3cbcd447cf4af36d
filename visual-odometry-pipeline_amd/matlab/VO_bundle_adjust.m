function [S,B] = VO_bundle_adjust(S,B,T_W_C,K,hyper_paras)
% Drop-in for Continuous_operation/VO_bundle_adjust.m.  The window struct B (main_demo.m:288-303) is kept
% by the library (vohip_window_*): keyframe bookkeeping and id look-ups on the host in C++, runBA and the
% re-refinement of the buffered normal frames on the GPU.  B passes through; only B.num_key and
% B.count_frame are kept current for callers that read them.
persistent started
if nargin < 5   % main_demo.m:410 calls with four arguments (is_BA is false there)
    hyper_paras = struct('min_depth', 1, 'max_depth', 100);
end
fresh = B.num_key == 1 && B.count_frame == 0 && B.num_normal_frame == 0;
if isempty(started) || fresh
    o = B.observation{1};
    n = o(1);
    uv = reshape(o(2:1+2*n), 2, n);   % main_demo.m:297-298 stores the first keyframe as (u, v)
    p = struct('window_size', B.window_size, 'keyframe_d', B.keyframe_d, ...
               'min_depth', hyper_paras.min_depth, 'max_depth', hyper_paras.max_depth);
    vohip_mex('window_create', K, p);
    vohip_mex('window_reset', B.landmarks(1:3,1:n), flipud(uv), twist2HomogMatrix(B.tau(:,1)));
    started = true;
end
keyframe = B.count_frame == B.keyframe_d;   % VO_bundle_adjust.m:3
[X, ~, offsets, poses, num_key] = vohip_mex('window_step', S.P, S.X, T_W_C);
S.X(1:3,:) = X;                                         % :62-70
for k = 1:numel(offsets)                                % :53-57, :80-84
    T = reshape(poses(:,k), 4, 4);
    S.est_rot(:,end+offsets(k)) = reshape(T(1:3,1:3), 9, 1);
    S.est_trans(:,end+offsets(k)) = T(1:3,4);
end
B.num_key = num_key;
if keyframe
    B.count_frame = 0;
else
    B.count_frame = B.count_frame + 1;
    B.num_normal_frame = min(B.num_normal_frame + 1, (B.window_size-1)*B.keyframe_d);
end
end
