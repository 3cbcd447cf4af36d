function [features, valid_key_points] = genKeypoints(image, method, options)
% Drop-in for utils/genKeypoints.m for the natively built detectors ('Harris', 'MinEigen', 'FAST').
% features is [] (the continuous-operation loop discards it); valid_key_points is a cornerPoints
% object, so .Location / .Metric / indexing behave as before.
opts = struct('MinQuality', 0.01, 'FilterSize', 5, 'MinContrast', 0.2);
if strcmp(method, 'FAST'), opts.MinQuality = 0.1; end   % detectFASTFeatures' own default
for k = 1:2:numel(options)
    opts.(options{k}) = options{k + 1};
end
switch method
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
