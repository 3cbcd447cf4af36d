function [features, valid_key_points] = genKeypoints(image, method, options)
% Drop-in for utils/genKeypoints.m for the natively built detectors ('SURF', 'Harris', 'MinEigen', 'FAST', 'BRISK').
% valid_key_points is a cornerPoints (SURF: SURFPoints) object, so .Location / .Metric / indexing behave as before.
% features is [] for the corner detectors (the continuous-operation loop discards it, main_demo.m:377) and, for
% SURF, the N x 64 descriptors of extractFeatures (genKeypoints.m:32), which the 'Des_match' bootstrap branch
% matches; the loop's own per-frame detections go through vohip.FrameLoop and never come here.
if nargin == 1, method = 'SURF'; end          % the reference's default
if nargin < 3, options = {}; end
opts = struct('MinQuality', 0.01, 'FilterSize', 5, 'MinContrast', 0.2, 'MetricThreshold', 1000, 'NumOctaves', 4);
if any(strcmp(method, {'FAST', 'BRISK'})), opts.MinQuality = 0.1; end   % detectFASTFeatures' / detectBRISKFeatures' own default
for k = 1:2:numel(options)
    opts.(options{k}) = options{k + 1};
end
switch method
    case 'SURF'
        [loc, metric, scale] = vohip_mex('detect_surf', image, opts.MetricThreshold, 16);
        [features, orientation] = vohip_mex('surf_features', image, loc, scale, 0);       % genKeypoints.m:32
        valid_key_points = SURFPoints(loc, 'Metric', metric, 'Scale', scale, 'Orientation', orientation);
        return
    case 'BRISK'                                                                        % genKeypoints.m:20
        [loc, metric, scale] = vohip_mex('detect_brisk', image, opts.MinContrast, opts.MinQuality, opts.NumOctaves, 16);
        features = [];      % (the binary BRISK descriptors are not built; the loop discards them, main_demo.m:377)
        valid_key_points = BRISKPoints(loc, 'Metric', metric, 'Scale', 12 * scale);   % (a BRISKPoints object's Scale is 12 at the image's own resolution)
        return
    case 'Harris'
        [loc, metric] = vohip_mex('detect_corners', image, opts.MinQuality, opts.FilterSize, 16, 0);
    case 'MinEigen'
        [loc, metric] = vohip_mex('detect_corners', image, opts.MinQuality, opts.FilterSize, 16, 1);
    case 'FAST'
        [loc, metric] = vohip_mex('detect_fast', image, opts.MinContrast, opts.MinQuality, 16);
    otherwise
        error('vohip:genKeypoints', 'method %s is not built natively', method);
end
features = [];
valid_key_points = cornerPoints(loc, 'Metric', metric);
end
